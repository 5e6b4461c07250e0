import sys, os, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import rel_l2
from helpers import hip_env_from_oracle, hip_mlp_from_net, reference_init_nets, to_device
from oracle import adp_oracle as orc
from gops_amd.utils.synthetic import act_dim_of, make_batch, obs_dim_of
from gops_amd import hip_backend as hb
dev = torch.device("cuda", 0)
for act in ("relu", "elu", "gelu", "tanh"):
  for hidden in ((256,256,256),(256,256)):
    cfg = dict(alg="INFADP", env_id="pyth_veh3dofconti", batch=300, horizon=6, pre_horizon=10, hidden=hidden, act=act, gamma=0.99)
    data = make_batch(cfg, 5)
    nets = reference_init_nets(cfg, 5, obs_dim_of(cfg), act_dim_of(cfg))
    env = orc.make_env(cfg["env_id"], pre_horizon=10)
    want = orc.infadp_pim_gradient(env, nets["policy"], nets["v_target"], data, cfg["horizon"], cfg["gamma"])
    out = {}
    for ss in ("1", "0"):
        os.environ["GOPS_SS"] = ss; os.environ["GOPS_SPLIT"] = "0"
        henv = hip_env_from_oracle(env, nets["policy"])
        pol, pw, pb = hip_mlp_from_net(nets["policy"], dev)
        vt = hip_mlp_from_net(nets["v_target"], dev)[0]
        B = 300
        ro = hb.Rollout(henv, pol, batch=B, horizon=6, gamma=0.99, finite_horizon=False, need_grad=True, value=vt)
        v = hb.lib().gops_rollout_variant(ctypes.byref(ro.desc))
        res = ro.forward(to_device(data, dev), want_final=True)
        gw, gb = [torch.empty_like(w) for w in pw], [torch.empty_like(b) for b in pb]
        ro.backward(torch.full((B,), -1.0 / B, device=dev), gw, gb)
        torch.cuda.synchronize()
        out[ss] = [t.cpu() for pair in zip(gw, gb) for t in pair]
        print(act, hidden, "ss", ss, "variant", v, "v_pi", f"{rel_l2(res['v_pi'].cpu(), want['v_pi']):.2e}", "per tensor", [f"{rel_l2(a, b):.1e}" for a, b in zip(out[ss], want["grads"])])
