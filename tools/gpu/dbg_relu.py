"""How far do the plane-split kernels land from the oracle for kinked activations at full size?"""
import sys, os, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import rel_l2
from helpers import hip_env_from_oracle, hip_mlp_from_net, reference_init_nets, to_device
from oracle import adp_oracle as orc
from gops_amd.utils.synthetic import act_dim_of, make_batch, obs_dim_of
from gops_amd import hip_backend as hb
torch.set_num_threads(16)
dev = torch.device("cuda", 0)
for act in ("relu", "selu", "elu"):
    cfg = dict(alg="FHADP", env_id="pyth_veh3dofconti", batch=4096, horizon=30, pre_horizon=30, hidden=(256, 256), act=act, gamma=1.0)
    data = make_batch(cfg, 0)
    nets = reference_init_nets(cfg, 0, obs_dim_of(cfg), act_dim_of(cfg))
    env = orc.make_env(cfg["env_id"], pre_horizon=30)
    want = orc.fhadp_gradient(env, nets["policy"], data, 30, 1.0)
    flat_ref = torch.cat([g.reshape(-1) for g in want["grads"]])
    for split in ("1", "0"):
        os.environ["GOPS_SPLIT"] = split
        henv = hip_env_from_oracle(env, nets["policy"])
        pol, pw, pb = hip_mlp_from_net(nets["policy"], dev)
        ro = hb.Rollout(henv, pol, batch=4096, horizon=30, gamma=1.0, finite_horizon=True, need_grad=True)
        v = hb.lib().gops_rollout_variant(ctypes.byref(ro.desc))
        res = ro.forward(to_device(data, dev))
        gw, gb = [torch.empty_like(w) for w in pw], [torch.empty_like(b) for b in pb]
        ro.backward(torch.full((4096,), -1.0 / 4096, device=dev), gw, gb)
        torch.cuda.synchronize()
        got = [t.cpu() for pair in zip(gw, gb) for t in pair]
        flat = torch.cat([t.reshape(-1) for t in got])
        print(act, "split", split, "variant", v, "flat", f"{rel_l2(flat, flat_ref):.2e}", "per tensor", [f"{rel_l2(a, b):.1e}" for a, b in zip(got, want["grads"])], flush=True)
