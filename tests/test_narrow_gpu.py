"""GPU: narrow policies on the plain streamed fp32 kernels keep their packed hidden-layer weights in LDS for the whole launch
(csrc/common.h gemm_layer_lds; the shapes of the reference's example scripts, e.g. example_train/fhadp/fhadp_mlp_idpendulum_serial.py:67-83:
MLP(64, 64) - which additionally runs kernel instantiations written out for exactly that shape, mlp_hidden_forward_n64 / mlp_backward_n64).  The LDS-resident path multiplies the same fragments in the same order as the path that streams them from L2 every
step (GOPS_VF_NO_NARROW_LDS): value, rewards and every gradient element must be IDENTICAL, and both match the oracle."""
import ctypes

import pytest
import torch

from conftest import rel_l2
from helpers import hip_env_from_oracle, hip_mlp_from_net, reference_init_nets, to_device
from oracle import adp_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-4   # north_star tolerance (fp32)

CASES = [
    # (env_id, extra cfg, hidden, act, batch, horizon)
    ("pyth_idpendulum", {}, (64, 64), "gelu", 40, 12),                       # ragged last tile; the example scripts' shape
    ("pyth_idpendulum", {}, (32, 48), "tanh", 16, 5),                        # unequal widths, one n-tile short of a full wave set
    ("pyth_veh3dofconti", dict(pre_horizon=10), (64, 64), "elu", 64, 10),    # 46 inputs -> three k-chunks in layer 0
    ("pyth_lq", dict(lq_config="s4a2"), (64, 64, 64), "relu", 48, 8),        # three hidden layers: 9216 floats, NOT resident (stays streamed)
    ("pyth_lq", dict(lq_config="s4a2"), (64, 32, 16), "relu", 48, 8),        # three hidden layers, resident
    ("pyth_idpendulum", {}, (64, 64), "elu", 1, 1),                          # one trajectory, one step (no input adjoint GEMM at t = 0)
    ("pyth_lq", dict(lq_config="s4a2"), (16,), "tanh", 33, 3),               # ONE hidden layer of one n-tile: three of four waves idle
]


def _run(cfg, flags, dev):
    from gops_amd import hip_backend as hb
    from gops_amd.utils.synthetic import act_dim_of, make_batch, obs_dim_of
    data = make_batch(cfg, 5)
    nets = reference_init_nets(cfg, 5, obs_dim_of(cfg), act_dim_of(cfg))
    env = orc.make_env(cfg["env_id"], **{k: cfg[k] for k in ("pre_horizon", "lq_config") if k in cfg})
    mlp, ws, bs = hip_mlp_from_net(nets["policy"], dev)
    ro = hb.Rollout(hip_env_from_oracle(env, nets["policy"]), mlp, batch=cfg["batch"], horizon=cfg["horizon"], gamma=cfg["gamma"],
                    finite_horizon=True, variant_flags=hb.DEFAULT_VARIANT_FLAGS | flags)
    res = ro.forward(to_device(data, dev))
    gw, gb = [torch.empty_like(w) for w in ws], [torch.empty_like(b) for b in bs]
    ro.backward(torch.full((cfg["batch"],), -1.0 / cfg["batch"], device=dev), gw, gb)
    torch.cuda.synchronize()
    grads = [t.clone() for pair in zip(gw, gb) for t in pair]
    return res["v_pi"].clone(), grads, (env, nets, data)


@pytest.mark.parametrize("env_id,extra,hidden,act,batch,horizon", CASES)
def test_lds_resident_weights_equal_streamed_weights_bit_for_bit(env_id, extra, hidden, act, batch, horizon):
    from gops_amd import hip_backend as hb
    dev = torch.device("cuda", 0)
    cfg = dict(alg="FHADP", env_id=env_id, batch=batch, horizon=horizon, hidden=hidden, act=act, gamma=0.99, **extra)
    cfg.setdefault("pre_horizon", horizon)
    v_a, g_a, (env, nets, data) = _run(cfg, 0, dev)
    # default (obs-64-64-act: the kernels written out for that shape), generic LDS-resident path, weights streamed from L2
    for flags in (hb.VF_NO_NARROW_N64, hb.VF_NO_NARROW_LDS):
        v_b, g_b, _ = _run(cfg, flags, dev)
        assert torch.equal(v_a, v_b), flags
        for a, b in zip(g_a, g_b):
            assert torch.equal(a, b), flags
    ref = orc.fhadp_gradient(env, nets["policy"], data, horizon, 0.99)
    got = torch.cat([t.reshape(-1).cpu() for t in g_a]).double()
    want = torch.cat([t.reshape(-1) for t in ref["grads"]]).double()
    assert rel_l2(got, want) < TOL
    assert abs(-v_a.double().mean().item() - ref["loss"].item()) < TOL * max(1.0, abs(ref["loss"].item()))
