"""Reproducibility hunt, round 4: run the SAME streamed-split forward launch several times with the GOPS_DUMP build
(libgops_hip_dump.so: every thread of the veh3dofconti streamed-split forward records 64 floats per step) and report, for
every tile whose records differ between two launches, the FIRST step and the record slots / threads that differ.

  GOPS_HIP_LIB=gops_amd/libgops_hip_dump.so GOPS_SS_VEH=1 GOPS_SSB=0 python tools/gpu/dbg_dump_diff.py [case] [runs]
"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import hip_env_from_oracle, hip_mlp_from_net, reference_init_nets, to_device
from oracle import adp_oracle as orc
from gops_amd import hip_backend as hb
from gops_amd.utils.synthetic import act_dim_of, make_batch, obs_dim_of

SLOTS = {0: "ya", 1: "steer", 2: "ax", 3: "dflag", **{4 + i: f"s{i}" for i in range(6)}, **{10 + i: f"sn{i}" for i in range(6)},
         16: "sin'", 17: "cos'", **{18 + i: f"rp{i}" for i in range(4)}, 22: "xtf", 23: "ytf", 24: "ptf", 25: "utf", 26: "rr", 27: "v_acc",
         28: "xs[c]", 29: "xs[c+16]", 30: "xs[c+32]", **{32 + i: f"h0[{i}]" for i in range(16)}, **{48 + i: f"hL[{i}]" for i in range(16)}}
dev = torch.device("cuda", 0)
_VEH = dict(alg="INFADP", env_id="pyth_veh3dofconti", batch=4800, horizon=4, pre_horizon=10, hidden=(256, 256), act="elu", gamma=0.99)
CASES = {"veh_p10": _VEH,
         "veh_fhadp_3x256": dict(alg="FHADP", env_id="pyth_veh3dofconti", batch=4800, horizon=4, pre_horizon=10, hidden=(256, 256, 256), act="relu", gamma=0.99),
         "veh_h8": dict(_VEH, horizon=8)}
name = sys.argv[1] if len(sys.argv) > 1 else "veh_p10"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = CASES[name]
data = make_batch(cfg, 5)
nets = reference_init_nets(cfg, 5, obs_dim_of(cfg), act_dim_of(cfg))
env = orc.make_env(cfg["env_id"], pre_horizon=cfg.get("pre_horizon", 10))
fh = cfg["alg"] == "FHADP"
B, H = data["obs"].shape[0], cfg["horizon"]
tiles = (B + 15) // 16
ddev = to_device(data, dev)
lib = hb.lib()
lib.gops_dbg_dump_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
n = tiles * H * 256 * 64
base = None
for it in range(runs):
    henv = hip_env_from_oracle(env, nets["policy"])
    pol, pw, pb = hip_mlp_from_net(nets["policy"], dev)
    vt = None if fh else hip_mlp_from_net(nets["v_target"], dev)[0]
    ro = hb.Rollout(henv, pol, batch=B, horizon=H, gamma=cfg["gamma"], finite_horizon=fh, need_grad=False, value=vt)
    ro.workspace.random_(0, 256)
    res = ro.forward(ddev, want_rewards=True, want_final=True)
    torch.cuda.synchronize()
    rec = np.empty(n, dtype=np.float32)
    rc = lib.gops_dbg_dump_read(rec.ctypes.data_as(ctypes.c_void_p), rec.nbytes)
    assert rc == 0, rc
    rec = rec.view(np.uint32).reshape(tiles, H, 256, 64)
    v = res["v_pi"].cpu().numpy().view(np.uint32)
    if base is None:
        base, base_v = rec, v
        print(name, "run 0 recorded; variant", lib.gops_rollout_variant(ctypes.byref(ro.desc)))
        del ro
        continue
    dv = np.nonzero(v != base_v)[0]
    diff = rec != base
    bad_tiles = np.nonzero(diff.any(axis=(1, 2, 3)))[0]
    print(f"run {it}: v_pi rows differing {dv.size} (tiles {sorted(set((dv // 16).tolist()))[:12]}); tiles with differing records: {bad_tiles.size} {bad_tiles[:12].tolist()}")
    for tl in bad_tiles[:6]:
        d = diff[tl]
        t0 = int(np.nonzero(d.any(axis=(1, 2)))[0][0])
        for t in range(t0, min(H, t0 + 2)):
            sl = np.nonzero(d[t].any(axis=0))[0]
            print(f"   tile {tl} step {t}{' (FIRST)' if t == t0 else ''}: slots", [SLOTS.get(int(s), str(int(s))) for s in sl])
            for s in sl[:10]:
                tids = np.nonzero(d[t][:, s])[0]
                a = rec[tl, t, tids[:4], s].view(np.float32); b = base[tl, t, tids[:4], s].view(np.float32)
                print(f"      {SLOTS.get(int(s), s)}: {tids.size} threads; waves {sorted(set((tids // 64).tolist()))} parts {sorted(set((tids // 16).tolist()))[:16]} m {sorted(set((tids % 16).tolist()))[:16]}"
                      f" e.g. tid {tids[:4].tolist()} now {a.tolist()} base {b.tolist()}")
    # hypotheses for the two first-deviation patterns (sn0 / sn4 in lanes 48..63 of a wave), from the recorded inputs
    f = lambda a: a.view(np.float32).astype(np.float64)
    for tl in bad_tiles[:10]:
        d = diff[tl]
        t0 = int(np.nonzero(d.any(axis=(1, 2)))[0][0])
        for slot in (10, 14):
            tids = np.nonzero(d[t0][:, slot])[0][:3]
            for tid in tids:
                r = f(rec[tl, t0, tid]); b = f(base[tl, t0, tid])
                x, y, phi, u, v, w = r[4:10]
                steer, ax = r[1], r[2]
                if slot == 10:
                    t0v = u * np.cos(phi) - v * np.sin(phi)
                    eff = (r[10] - x) / 0.1
                    print(f"      sn0 tile {tl} t {t0} tid {tid}: base {b[10]:.6f} recomputed {x + 0.1 * t0v:.6f} wrong {r[10]:.6f}; t0 {t0v:.5f} effective t0 {eff:.5f} (diff {eff - t0v:+.5f}); "
                          f"sphi*v {np.sin(phi) * v:+.5f} cphi*u {np.cos(phi) * u:.5f} v' {b[14]:+.5f} w' {b[15]:+.5f} u {u:.4f} v {v:+.5f} y {y:.4f}")
                else:
                    den = 1412.0 * u + 0.1 * (128915.5 + 85943.6)
                    print(f"      sn4 tile {tl} t {t0} tid {tid}: base {b[14]:+.6f} wrong {r[14]:+.6f} ratio {r[14] / b[14]:.5f}; den_v/den_v(m:=0) = {den / (0.1 * (128915.5 + 85943.6)):.5f} (u {u:.4f})")
    del ro
