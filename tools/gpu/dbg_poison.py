"""Reproducibility / uninitialised-read hunt (DESIGN_LOG.md, round 4): the same streamed-split launch pair with the workspace poisoned (NaN bytes / random bytes)
before the forward; every result must be finite and bit-identical to the clean run."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from helpers import hip_env_from_oracle, hip_mlp_from_net, reference_init_nets, to_device
from oracle import adp_oracle as orc
from gops_amd import hip_backend as hb
from gops_amd.utils.synthetic import act_dim_of, make_batch, obs_dim_of

dev = torch.device("cuda", 0)
NEED_GRAD = os.environ.get("DBG_NOGRAD") is None   # DBG_NOGRAD=1: forward only, without the activation stash
_VEH = dict(alg="INFADP", env_id="pyth_veh3dofconti", batch=4800, horizon=4, pre_horizon=10, hidden=(256, 256), act="elu", gamma=0.99)
ALL_CASES = {
    # the open defect (run with GOPS_SS_VEH=1; GOPS_SSB=0 isolates the forward): 300 tiles, two workgroups on 44 CUs
    "veh_p10": _VEH,
    "veh_p5": dict(_VEH, pre_horizon=5),
    "veh_fhadp_3x256": dict(alg="FHADP", env_id="pyth_veh3dofconti", batch=4800, horizon=4, pre_horizon=10, hidden=(256, 256, 256), act="relu", gamma=0.99),
    "veh_h1": dict(_VEH, horizon=1),            # bit-stable
    "veh_256_tiles": dict(_VEH, batch=4096),    # bit-stable: one workgroup per CU
    # kinds that are on by default: bit-stable returns, gradient spread <= 1.4e-6 from the sweep
    "lq_many": dict(alg="INFADP", env_id="pyth_lq", lq_config="s4a2", batch=4800, horizon=6, hidden=(256, 256), act="gelu", gamma=0.99),
    "veh2_many": dict(alg="INFADP", env_id="pyth_veh2dofconti", batch=4800, horizon=4, pre_horizon=10, hidden=(256, 256), act="elu", gamma=0.99),
    "idp_many": dict(alg="FHADP", env_id="pyth_idpendulum", batch=4800, horizon=4, hidden=(256, 256, 256), act="elu", gamma=0.99),
    # register-stationary plane-split kernels (one workgroup per CU): bit-stable
    "veh_split": dict(alg="FHADP", env_id="pyth_veh3dofconti", batch=200, horizon=12, pre_horizon=30, hidden=(256, 256), act="elu", gamma=0.99),
}
# usage: [GOPS_SS_VEH=1] [GOPS_SSB=0] python tools/gpu/dbg_poison.py [case ...]     (default: veh_p10 lq_many)
CASES = {k: ALL_CASES[k] for k in (sys.argv[1:] or ["veh_p10", "lq_many"])}
for name, cfg in CASES.items():
    data = make_batch(cfg, 5)
    nets = reference_init_nets(cfg, 5, obs_dim_of(cfg), act_dim_of(cfg))
    env = orc.make_env(cfg["env_id"], pre_horizon=cfg.get("pre_horizon", 10), lq_config=cfg.get("lq_config", "s4a2"))
    fh = cfg["alg"] == "FHADP"
    B = data["obs"].shape[0]
    ddev = to_device(data, dev)
    base = None
    for mode in ["clean", "nan", "rand", "clean", "rand", "nan", "rand", "rand", "rand", "rand"]:
        henv = hip_env_from_oracle(env, nets["policy"])
        pol, pw, pb = hip_mlp_from_net(nets["policy"], dev)
        vt = None if fh else hip_mlp_from_net(nets["v_target"], dev)[0]
        ro = hb.Rollout(henv, pol, batch=B, horizon=cfg["horizon"], gamma=cfg["gamma"], finite_horizon=fh, need_grad=NEED_GRAD, value=vt)
        if mode == "nan": ro.workspace.fill_(255)           # 0xffffffff: NaN
        elif mode == "rand": ro.workspace.random_(0, 256)
        else: ro.workspace.zero_()
        res = ro.forward(ddev, want_rewards=True, want_final=True)
        gw, gb = [torch.empty_like(w) for w in pw], [torch.empty_like(b) for b in pb]
        if NEED_GRAD: ro.backward(torch.full((B,), -1.0 / B, device=dev), gw, gb)
        else: [t.zero_() for t in gw + gb]
        torch.cuda.synchronize()
        flat = torch.cat([res["v_pi"].reshape(-1)] + [t.reshape(-1) for pair in zip(gw, gb) for t in pair]).cpu()
        rw = res['rewards'].cpu(); fo = res['final_obs'].cpu()
        if base is None: base = flat; base_rw = rw; base_fo = fo
        same = torch.equal(flat, base)
        print("touch", os.environ.get("GOPS_TOUCH", "-"), "ssb", os.environ.get("GOPS_SSB", "1"), "nofuse", os.environ.get("GOPS_NO_FUSED_DWOUT", "-"), "spec", os.environ.get("GOPS_DW_SPEC", "1"), "exact", os.environ.get("GOPS_DW_EXACT", "-"), name, mode, "finite", bool(torch.isfinite(flat).all()), "bit-identical", same,
              "" if same else f"rel-L2 of the gradient {float((flat[B:] - base[B:]).norm() / base[B:].norm()):.3e} max rel diff {float(((flat - base).abs() / (base.abs() + 1e-12)).max()):.3e} v_pi same {torch.equal(flat[:B], base[:B])}")
        if not same:
            for t in range(rw.shape[0]):
                dd = (rw[t] != base_rw[t]).nonzero().reshape(-1)
                print("    step", t, "rewards differing:", dd.numel(), "tiles", sorted(set((dd // 16).tolist()))[:8], "max abs", float((rw[t] - base_rw[t]).abs().max()))
            dd = (fo != base_fo).any(dim=1).nonzero().reshape(-1)
            cols = (fo != base_fo).any(dim=0).nonzero().reshape(-1).tolist()
            print("    final_obs rows differing:", dd.numel(), "columns", cols[:20], "max abs", float((fo - base_fo).abs().max()))
            d = (flat[:B] != base[:B]).nonzero().reshape(-1)
            print("    differing v_pi:", d.numel(), "rows; tiles", sorted(set((d // 16).tolist()))[:20], "max abs", float((flat[:B] - base[:B]).abs().max()), "of", float(base[:B].abs().max()))
        del ro
