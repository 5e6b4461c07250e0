"""cfg3 (veh3dof INFADP, 256^3 relu, B = 8192): distance of the policy gradient from the reference fixture - default kernels (plane-split step\nloop + exact-fp32 tail value net, plane-split sweep) and the exact fp32-MFMA kernels.   python tools/gpu/cfg3_parity_margin.py"""
import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from helpers import *
from gops_amd import hip_backend as hb
from gops_amd.utils.synthetic import CONFIGS, make_batch, obs_dim_of, act_dim_of
import test_hip_parity as T
rel_l2 = T.rel_l2
name = "cfg3_veh3dof_infadp_b8192"
cfg = CONFIGS[name]; g = T.load_golden("big_" + name); dev = torch.device("cuda", 0)
data = make_batch(cfg, 0); nets = reference_init_nets(cfg, 0, obs_dim_of(cfg), act_dim_of(cfg))
env = T.oracle_env(cfg, {}, g); henv = hip_env_from_oracle(env, nets["policy"]); ddev = to_device(data, dev); B = cfg["batch"]
pol, pw, pb = hip_mlp_from_net(nets["policy"], dev); vt, _, _ = hip_mlp_from_net(nets["v_target"], dev)
for flags, label in ((None, "default (plane-split step loop, exact tail)"), (hb.VF_NO_STREAMED_SPLIT_FWD | hb.VF_NO_STREAMED_SPLIT_BWD, "exact fp32 kernels")):
    kw = {} if flags is None else dict(variant_flags=flags)
    ro2 = hb.Rollout(henv, pol, batch=B, horizon=cfg["horizon"], gamma=cfg["gamma"], finite_horizon=False, need_grad=True, value=vt, **kw)
    res = ro2.forward(ddev)
    gw, gb = [torch.empty_like(w) for w in pw], [torch.empty_like(b) for b in pb]
    ro2.backward(torch.full((B,), -1.0 / B, device=dev), gw, gb)
    torch.cuda.synchronize()
    grads = [t for pair in zip(gw, gb) for t in pair]
    errs = [rel_l2(gr.reshape(-1).cpu()[torch.from_numpy(g[f"pim_grad/idx{i}"])], g[f"pim_grad/val{i}"]) for i, gr in enumerate(grads)]
    print(label, "loss rel", abs(-res["v_pi"].double().mean().item() - float(g["pim_loss"])) / abs(float(g["pim_loss"])), "max grad rel_l2", max(errs))
