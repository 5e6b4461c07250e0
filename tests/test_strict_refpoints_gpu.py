"""GPU: `strict_reference_points=True` - the appended reference points of the veh3dofconti / veh2dofconti rollouts come from the
host's torch CPU ops (gops_amd/env/env_ocp/resources/ref_traj_host.py), i.e. they are the values the reference itself computes
on this host, and everything the 1 ms finite-difference heading reaches meets the reference's own tolerances with NO exemption:

* single steps through `create_env_model(...).forward`: every observation element at rtol 1e-5 (tests/test_hip_parity.py
  `test_env_step_vs_reference_fixture` needs an outlier share for the kernels' own headings);
* the gradients of the reference-TRAINED 256-wide veh3dofconti networks through `create_alg(...)` at 1e-4 with nothing handed in
  by the test (tests/test_trained256_gpu.py hands the oracle's points in for its 1e-4 assertion)."""
import numpy as np
import pytest
import torch

from conftest import golden_meta, load_golden, rel_l2
from helpers import INFO_KEYS, data_from_golden, to_device
from test_alg_gpu import _kwargs

from gops_amd.create_pkg.create_alg import create_alg
from gops_amd.create_pkg.create_env_model import create_env_model

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


@pytest.mark.parametrize("name", ["step_veh_p10", "step_veh_p30", "step_veh_p10_refpara", "step_veh_p10_nomask", "step_veh2dof_p10",
                                  "step_veh2dof_p10_refpara"])
def test_env_model_forward_is_elementwise_at_the_references_tolerance(name, dev):
    g = load_golden(name)
    meta = golden_meta(g)
    cfg, extra = meta["cfg"], dict(meta["extra"])
    model = create_env_model(**cfg, **extra, use_gpu=True, strict_reference_points=True)   # (the wrapper chain the fixture was recorded with)
    assert model.strict_reference_points
    data = to_device(data_from_golden(g), dev)
    obs, done = data["obs"], data["done"]
    info = {k: data[k] for k in INFO_KEYS if k in data}
    for s in range(int(g["meta/nsteps"])):
        a = torch.from_numpy(g[f"s{s}/act"]).to(dev)
        obs, r, done, info = model.forward(obs, a, done, info)
        np.testing.assert_allclose(obs.cpu().numpy(), g[f"s{s}/obs"], rtol=1e-5, atol=2e-5)      # EVERY element, no outlier share
        np.testing.assert_allclose(r.cpu().numpy(), g[f"s{s}/rew"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(info["state"].cpu().numpy(), g[f"s{s}/state"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(info["ref_points"][:, -1].cpu().numpy(), g[f"s{s}/ref_last"], rtol=1e-5, atol=2e-6)   # heading included
        assert np.array_equal(done.cpu().numpy() != 0, np.asarray(g[f"s{s}/done"]) != 0)
        info = {k: v for k, v in info.items() if v is not None}


def _load(name, **more):
    g = load_golden(name)
    meta = golden_meta(g)
    cfg = meta["cfg"]
    alg = create_alg(**dict(_kwargs(cfg, meta["extra"], meta["seed"]), **more))
    alg.load_state_dict({k[3:]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith("sd/")})
    alg.networks.cuda()
    return alg, g, cfg


def _flat(ts):
    return torch.cat([t.detach().reshape(-1).double().cpu() for t in ts])


def _golden_flat(g, prefix):
    n = len([k for k in g if k.startswith(prefix)])
    return torch.cat([torch.from_numpy(np.array(g[f"{prefix}{i}"])).reshape(-1).double() for i in range(n)])


def test_trained_fhadp_veh_policy_through_create_alg(dev):
    """`t256_fhadp_veh_p30_elu` (the TARGET's model, trained by the unmodified reference): 1e-4 with the product's own strict mode."""
    alg, g, cfg = _load("t256_fhadp_veh_p30_elu", strict_reference_points=True, precision_check_interval=0)
    alg.gamma = cfg["gamma"]
    data = data_from_golden(g)
    assert "ref_appended" not in data
    tb, info = alg.get_remote_update_info(data, 0)
    err = rel_l2(_flat(info["grad"]), _golden_flat(g, "grad/"))
    worst = max(rel_l2(gr.cpu(), g[f"grad/{i}"]) for i, gr in enumerate(info["grad"]))
    loss = float(tb["Loss/Actor loss-RL iter"])
    print(f"t256_fhadp_veh_p30_elu strict: grad {err:.2e} (worst tensor {worst:.2e}), loss {abs(loss - float(g['loss'])):.1e}")
    assert err < TOL and worst < TOL
    assert abs(loss - float(g["loss"])) <= TOL * max(1.0, abs(float(g["loss"])))
    # the same call without the mode: the kernels' own headings (bounded, not at the bar on every host - tests/test_trained256_gpu.py)
    alg2, _, _ = _load("t256_fhadp_veh_p30_elu", precision_check_interval=0)
    alg2.gamma = cfg["gamma"]
    _, info2 = alg2.get_remote_update_info(data, 0)
    print(f"  default (kernel headings): grad {rel_l2(_flat(info2['grad']), _golden_flat(g, 'grad/')):.2e}")


def test_trained_infadp_veh_networks_through_create_alg(dev):
    """`t256_infadp_veh_p10_relu3` (cfg3's shape; 1.6e-4 with the kernels' headings): PEV and PIM at 1e-4 in strict mode."""
    alg, g, cfg = _load("t256_infadp_veh_p10_relu3", strict_reference_points=True, precision_check_interval=0)
    alg.gamma, alg.forward_step = cfg["gamma"], cfg["horizon"]
    data = data_from_golden(g)
    tb, info = alg.get_remote_update_info(data, 0)       # PEV
    e_v = rel_l2(_flat(info["v"]), _golden_flat(g, "pev_grad/"))
    assert abs(float(tb["Loss/Critic loss-RL iter"]) - float(g["pev_loss"])) <= TOL * max(1.0, abs(float(g["pev_loss"])))
    tb, info = alg.get_remote_update_info(data, 1)       # PIM
    e_p = rel_l2(_flat(info["policy"]), _golden_flat(g, "pim_grad/"))
    worst = max(rel_l2(gr.cpu(), g[f"pim_grad/{i}"]) for i, gr in enumerate(info["policy"]))
    print(f"t256_infadp_veh_p10_relu3 strict: PEV grad {e_v:.2e}, PIM grad {e_p:.2e} (worst tensor {worst:.2e})")
    assert e_v < TOL and e_p < TOL and worst < TOL
    assert abs(float(tb["Loss/Actor loss-RL iter"]) - float(g["pim_loss"])) <= TOL * max(1.0, abs(float(g["pim_loss"])))


def test_prefetched_points_equal_the_ones_computed_on_the_spot_and_updates_replay(dev):
    """`prefetch_reference_points(next_batch)` evaluates on the side thread while the GPU works; the update that receives the batch
    collects them.  Same weights after 6 updates as with on-the-spot evaluation, also once the update replays as a HIP graph."""
    from gops_amd.utils.synthetic import make_batch
    cfg = dict(alg="FHADP", env_id="pyth_veh3dofconti", batch=64, horizon=10, pre_horizon=10, hidden=(64, 64), act="elu", gamma=1.0)
    batches = [make_batch(cfg, 40 + i) for i in range(6)]

    sd0 = {k: v.clone() for k, v in create_alg(**_kwargs(cfg, {}, 3)).state_dict().items()}

    def run(prefetch):
        alg = create_alg(**dict(_kwargs(cfg, {}, 3), strict_reference_points=True))
        alg.load_state_dict(sd0)   # (the factories draw fresh initial weights per call)
        alg.networks.cuda()
        dbs = [to_device(b, dev) for b in batches]
        if prefetch:
            alg.prefetch_reference_points(dbs[0])
        for i, b in enumerate(dbs):
            if prefetch and i + 1 < len(dbs):
                alg.prefetch_reference_points(dbs[i + 1])
            alg.local_update(b, i)
        torch.cuda.synchronize()
        assert alg._update_graph.graph is not None and not alg._update_graph.failed   # (B x H = 640: the update replays as a graph)
        return [p.detach().clone() for p in alg.networks.policy.parameters()], alg._reference_pipeline().evaluated

    w_a, n_a = run(False)
    w_a2, _ = run(False)
    assert all(torch.equal(a, b) for a, b in zip(w_a, w_a2)), "two identical runs differ"
    w_b, n_b = run(True)
    assert n_a == n_b == len(batches)
    assert all(torch.equal(a, b) for a, b in zip(w_a, w_b))
    # ... and the mode changes something: the default kernels' headings give (slightly) different weights
    alg = create_alg(**_kwargs(cfg, {}, 3))
    alg.networks.cuda()
    assert alg._reference_pipeline() is None


@pytest.mark.parametrize("algname", ["INFADP", "FHADP2"])
def test_strict_mode_equals_handing_the_same_points_in(algname, dev):
    """The mode is plumbing around `ref_appended`: an algorithm in strict mode and the same algorithm fed the provider's table by
    hand produce identical gradients (INFADP: both modes of an update pair; FHADP2: the open-loop rollout)."""
    from gops_amd.env.env_ocp.resources.ref_traj_host import HostRefTraj
    from gops_amd.utils.synthetic import make_batch
    if algname == "INFADP":
        cfg = dict(alg="INFADP", env_id="pyth_veh3dofconti", batch=96, horizon=6, pre_horizon=10, hidden=(64, 64), act="relu", gamma=0.99)
        extra = {}
    else:
        cfg = dict(alg="FHADP2", env_id="pyth_veh3dofconti", batch=96, horizon=8, pre_horizon=8, hidden=(64, 64), act="elu", gamma=1.0)
        extra = dict(policy_func_name="FiniteHorizonFullPolicy")
    data = to_device(make_batch(dict(cfg, alg="FHADP" if algname == "FHADP2" else cfg["alg"]), 5), dev)
    kw = dict(_kwargs(dict(cfg, alg="FHADP" if algname == "FHADP2" else "INFADP"), {}, 9), algorithm=algname, **extra)
    sd0 = None
    grads = {}
    for mode in ("strict", "by_hand"):
        alg = create_alg(**dict(kw, strict_reference_points=(mode == "strict")))
        if sd0 is None:
            sd0 = {k: v.clone() for k, v in alg.state_dict().items()}
        alg.load_state_dict(sd0)
        alg.networks.cuda()
        d = dict(data)
        if mode == "by_hand":
            H = cfg["horizon"]
            if algname == "INFADP":
                alg.forward_step = H
            d["ref_appended"] = HostRefTraj().appended_points(data["ref_time"], data["path_num"], data["u_num"], H, cfg["pre_horizon"]).to(dev)
        elif algname == "INFADP":
            alg.forward_step = cfg["horizon"]
        if algname == "INFADP":
            alg.gamma = cfg["gamma"]
            out = []
            for it in (0, 1):
                _, info = alg.get_remote_update_info(d, it)
                out += [g.clone() for g in list(info.values())[0]]
            grads[mode] = out
        else:
            alg.gamma = cfg["gamma"]
            _, info = alg.get_remote_update_info(d, 0)
            grads[mode] = [g.clone() for g in info["grad"]]
    assert all(torch.equal(a, b) for a, b in zip(grads["strict"], grads["by_hand"]))
    assert any(g.abs().max() > 0 for g in grads["strict"])
