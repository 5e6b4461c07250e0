"""GPU: the pyth_mobilerobot model (GOPS_ENV_MOBILEROBOT kernels) against the reference's fixtures and the oracle - env steps,
SPIL / FHADPExterior / INFADP gradients through the C ABI, the algorithm classes, adjoint I/O.  The obstacle robot's
np.random.normal draws recorded from the reference run are replayed through GopsRolloutIn.noise / GopsStepIO.noise."""
import numpy as np
import pytest
import torch

from conftest import golden_meta, load_golden, rel_l2
from helpers import data_from_golden, hip_env_from_oracle, hip_mlp_from_net, nets_from_golden, oracle_env, to_device
from oracle import adp_oracle as orc

from gops_amd import hip_backend as hb

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _grads_close(gw, gb, g, prefix, what):
    k = 0
    for w_, b_ in zip(gw, gb):
        for t in (w_, b_):
            err = rel_l2(t.cpu(), g[f"{prefix}/{k}"])
            assert err < TOL, (what, k, err)
            k += 1


def test_env_step_vs_reference_fixture(dev):
    g = load_golden("step_mobilerobot")
    env = hip_env_from_oracle(oracle_env(golden_meta(g)["cfg"], {}, g))
    obs, done = torch.from_numpy(g["in/obs"]).to(dev), torch.from_numpy(g["in/done"]).to(dev)
    for s in range(int(g["meta/nsteps"])):
        info = dict(noise=torch.from_numpy(g[f"s{s}/noise"]).to(dev))
        obs, r, done, info = hb.env_step(env, obs, torch.from_numpy(g[f"s{s}/act"]).to(dev), done, info)
        np.testing.assert_allclose(obs.cpu().numpy(), g[f"s{s}/obs"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(r.cpu().numpy(), g[f"s{s}/rew"], rtol=1e-5, atol=1e-5)
        assert np.array_equal(done.cpu().numpy() != 0, g[f"s{s}/done"])
        np.testing.assert_allclose(info["constraint"].cpu().numpy(), g[f"s{s}/constraint"], rtol=1e-5, atol=1e-5)


def test_spil_rollouts_vs_reference_fixture(dev):
    """SPIL's two rollouts through the C ABI: PEV (no-grad, unmasked tail value, safe flags) and PIM (gradient through the
    return and the Phi-products), each with the draws the reference used."""
    g = load_golden("spil_mobilerobot")
    cfg = golden_meta(g)["cfg"]
    env = oracle_env(cfg, {}, g)
    nets, _ = nets_from_golden(g, cfg)
    data = data_from_golden(g)
    ddev = to_device(data, dev)
    B, H = data["obs"].shape[0], cfg["horizon"]
    henv = hip_env_from_oracle(env, nets["policy"])
    pol, pw, pb = hip_mlp_from_net(nets["policy"], dev)
    vt, _, _ = hip_mlp_from_net(nets["v_target"], dev)
    ro = hb.Rollout(henv, pol, batch=B, horizon=H, gamma=cfg["gamma"], finite_horizon=False, need_grad=False, value=vt,
                    tail_unmasked=True)
    res = ro.forward(dict(ddev, noise=ddev["noise_pev"]))
    ref = orc.spil_pev(env, nets["policy"], nets["v"], nets["v_target"], dict(data, noise=data["noise_pev"]), H, cfg["gamma"])
    np.testing.assert_allclose(res["constraint_prods"][1].mean().item(), ref["safe_prob"].item(), atol=1e-6)
    np.testing.assert_allclose(res["constraint_prods"][1].mean().item(), float(g["safe_prob"][0]), atol=1e-6)
    # PIM
    lam = g["lam"]
    w_r, w_c = 1 / (1 + lam.sum()), lam / (1 + lam.sum())
    ro2 = hb.Rollout(henv, pol, batch=B, horizon=H, gamma=cfg["gamma"], finite_horizon=False, need_grad=True)
    res = ro2.forward(dict(ddev, noise=ddev["noise_pim"]))
    pim = orc.spil_pim(env, nets["policy"], dict(data, noise=data["noise_pim"]), H, cfg["gamma"], w_r, w_c)
    assert rel_l2(res["v_pi"].cpu(), pim["r_sum"]) < TOL
    assert rel_l2(res["constraint_prods"][:1].t().cpu(), pim["c_mul"]) < TOL
    w_c_t = torch.tensor(np.asarray(w_c, dtype=np.float32), device=dev)
    c_mul = res["constraint_prods"][:1]
    gw, gb = [torch.empty_like(w) for w in pw], [torch.empty_like(b) for b in pb]
    ro2.backward(torch.full((B,), -float(w_r) / B, device=dev), gw, gb, grad_constraint_prod=(-(w_c_t[:, None] / B) * c_mul).contiguous())
    torch.cuda.synchronize()
    _grads_close(gw, gb, g, "pim_grad", "spil pim")


def test_constraint_sums_gradient_vs_reference_fixture(dev):
    """FHADPExterior on pyth_mobilerobot: loss = -mean(v) + penalty * mean(sum_t gamma^t max(c_t, 0)^2) - the constraint-sum
    outputs and their adjoint."""
    g = load_golden("fhadp_ext_mobilerobot")
    meta = golden_meta(g)
    cfg = meta["cfg"]
    env = oracle_env(cfg, {}, g)
    nets, _ = nets_from_golden(g, cfg)
    data = data_from_golden(g)
    B, H, pen = data["obs"].shape[0], cfg["horizon"], meta["extra"]["penalty"]
    henv = hip_env_from_oracle(env, nets["policy"])
    pol, pw, pb = hip_mlp_from_net(nets["policy"], dev)
    ro = hb.Rollout(henv, pol, batch=B, horizon=H, gamma=cfg["gamma"], finite_horizon=True)
    res = ro.forward(to_device(data, dev))
    loss = (-res["v_pi"].double().mean() + pen * res["constraint_sums"][0].double().mean()).item()
    assert abs(loss - float(g["loss"])) <= TOL * max(1.0, abs(float(g["loss"])))
    assert abs(res["constraint_sums"][0].mean().item() - float(g["tb/Loss/Actor constraint loss-RL iter"])) <= TOL   # (logged without the penalty factor)
    gc = torch.zeros(3, B, device=dev)
    gc[0] = pen / B
    gw, gb = [torch.empty_like(w) for w in pw], [torch.empty_like(b) for b in pb]
    ro.backward(torch.full((B,), -1.0 / B, device=dev), gw, gb, grad_constraint=gc)
    torch.cuda.synchronize()
    _grads_close(gw, gb, g, "grad", "fhadp exterior")


def test_infadp_vs_reference_fixture(dev):
    g = load_golden("infadp_mobilerobot_gelu")
    cfg = golden_meta(g)["cfg"]
    env = oracle_env(cfg, {}, g)
    nets, _ = nets_from_golden(g, cfg)
    data = data_from_golden(g)
    ddev = to_device(data, dev)
    B, H = data["obs"].shape[0], cfg["horizon"]
    henv = hip_env_from_oracle(env, nets["policy"])
    pol, pw, pb = hip_mlp_from_net(nets["policy"], dev)
    vt, _, _ = hip_mlp_from_net(nets["v_target"], dev)
    v, vw, vb = hip_mlp_from_net(nets["v"], dev)
    ro = hb.Rollout(henv, pol, batch=B, horizon=H, gamma=cfg["gamma"], finite_horizon=False, need_grad=False, value=vt)
    backup = ro.forward(dict(ddev, noise=ddev["noise_pev"]))["v_pi"]
    vn = hb.ValueNet(v, B)
    vo = vn.forward(ddev["obs"])
    assert abs(((vo - backup) ** 2).mean().item() - float(g["pev_loss"])) <= TOL * max(1.0, abs(float(g["pev_loss"])))
    gw, gb = [torch.empty_like(w) for w in vw], [torch.empty_like(b) for b in vb]
    vn.backward(ddev["obs"], (2.0 / B) * (vo - backup), gw, gb)
    torch.cuda.synchronize()
    _grads_close(gw, gb, g, "pev_grad", "infadp pev")
    ro2 = hb.Rollout(henv, pol, batch=B, horizon=H, gamma=cfg["gamma"], finite_horizon=False, need_grad=True, value=vt)
    res = ro2.forward(dict(ddev, noise=ddev["noise_pim"]), want_final=True)
    ref = orc.infadp_pim_gradient(env, nets["policy"], nets["v_target"], dict(data, noise=data["noise_pim"]), H, cfg["gamma"])
    assert rel_l2(res["final_obs"].cpu(), ref["final_obs"]) < TOL
    assert np.array_equal(res["final_done"].cpu().numpy() != 0, ref["final_done"].numpy())
    gw, gb = [torch.empty_like(w) for w in pw], [torch.empty_like(b) for b in pb]
    ro2.backward(torch.full((B,), -1.0 / B, device=dev), gw, gb)
    torch.cuda.synchronize()
    assert abs(-res["v_pi"].double().mean().item() - float(g["pim_loss"])) <= TOL * max(1.0, abs(float(g["pim_loss"])))
    _grads_close(gw, gb, g, "pim_grad", "infadp pim")


@pytest.mark.parametrize("batch,horizon,act", [(1, 3, "tanh"), (17, 12, "elu"), (100, 25, "relu")])
def test_ragged_batches_and_long_horizons_match_oracle(batch, horizon, act, dev):
    """Synthetic batches (a third of the obstacles start next to the ego robot), noise drawn here: return, products, sums and
    gradient of a mixed loss against the oracle's autograd."""
    from gops_amd.utils.synthetic import make_batch
    torch.manual_seed(batch)
    env = orc.make_env("pyth_mobilerobot")
    pol = orc.make_net([13, 64, 64, 2], act, seed=batch, act_high=torch.ones(2), act_low=-torch.ones(2))
    data = make_batch(dict(env_id="pyth_mobilerobot", batch=batch), seed=batch + 1)
    data["done"][::5] = 1.0
    data["noise"] = torch.randn(horizon, batch, 2) * torch.tensor(hb.MOBILEROBOT_NOISE_STD)
    gamma = 0.97
    # oracle: loss = -mean(v) + 0.7 mean(sum gamma^t max(c, 0)^2) - 0.3 mean(prod Phi(c))
    o, d, info = data["obs"], data["done"], data
    v, ext, mul = torch.zeros(batch), torch.zeros(batch), torch.ones(batch)
    for t in range(horizon):
        a = orc.policy_forward(pol, o, None)
        o, r, d, info = orc.env_forward(env, o, a, d, info)
        c = info["constraint"][:, 0]
        v = v + gamma ** t * r
        ext = ext + gamma ** t * torch.clamp(c, min=0) ** 2
        mul = mul * orc.spil_phi(c)
    loss = -v.mean() + 0.7 * ext.mean() - 0.3 * mul.mean()
    ref_grads = orc._grads(loss, pol)
    henv = hip_env_from_oracle(env, pol)
    mlp, pw, pb = hip_mlp_from_net(pol, dev)
    ro = hb.Rollout(henv, mlp, batch=batch, horizon=horizon, gamma=gamma, finite_horizon=False)
    res = ro.forward(to_device(data, dev), want_final=True)
    assert rel_l2(res["v_pi"].cpu(), v.detach()) < TOL
    assert rel_l2(res["constraint_sums"][0].cpu(), ext.detach()) < TOL or ext.abs().max() < 1e-6
    assert rel_l2(res["constraint_prods"][0].cpu(), mul.detach()) < TOL
    assert rel_l2(res["final_obs"].cpu(), o.detach()) < TOL
    gc = torch.zeros(3, batch, device=dev)
    gc[0] = 0.7 / batch
    gw, gb = [torch.empty_like(w) for w in pw], [torch.empty_like(b) for b in pb]
    ro.backward(torch.full((batch,), -1.0 / batch, device=dev), gw, gb, grad_constraint=gc,
                grad_constraint_prod=((-0.3 / batch) * res["constraint_prods"][:1]).contiguous())
    torch.cuda.synchronize()
    got = [t for pair in zip(gw, gb) for t in pair]
    flat, flat_ref = torch.cat([x.reshape(-1).cpu() for x in got]), torch.cat([x.reshape(-1) for x in ref_grads])
    assert rel_l2(flat, flat_ref) < TOL, rel_l2(flat, flat_ref)


def test_observation_adjoint_matches_autograd(dev):
    """gops_rollout_backward_adj on pyth_mobilerobot: d(loss)/d(initial observation) with a terminal term on the final
    observation (the obstacle columns only reach the result through this adjoint)."""
    from gops_amd.utils.synthetic import make_batch
    B, H, gamma = 37, 6, 0.95
    torch.manual_seed(3)
    env = orc.make_env("pyth_mobilerobot")
    pol = orc.make_net([13, 32, 32, 2], "tanh", seed=5, act_high=torch.ones(2), act_low=-torch.ones(2))
    data = make_batch(dict(env_id="pyth_mobilerobot", batch=B), seed=8)
    data["noise"] = torch.randn(H, B, 2) * torch.tensor(hb.MOBILEROBOT_NOISE_STD)
    gfo = torch.randn(B, 13) * 0.1
    obs0 = data["obs"].clone().requires_grad_(True)
    o, d, info = obs0, data["done"], data
    v = torch.zeros(B)
    for t in range(H):
        o, r, d, info = orc.env_forward(env, o, orc.policy_forward(pol, o, None), d, info)
        v = v + gamma ** t * r
    loss = -v.mean() + (o * gfo).sum()
    params = [p for pair in zip(pol["w"], pol["b"]) for p in pair]
    grads = torch.autograd.grad(loss, [obs0] + params)
    henv = hip_env_from_oracle(env, pol)
    mlp, pw, pb = hip_mlp_from_net(pol, dev)
    ro = hb.Rollout(henv, mlp, batch=B, horizon=H, gamma=gamma, finite_horizon=False)
    ro.forward(to_device(data, dev))
    gw, gb = [torch.empty_like(w) for w in pw], [torch.empty_like(b) for b in pb]
    gobs = ro.backward_adj(torch.full((B,), -1.0 / B, device=dev), gw, gb, grad_final_obs=gfo.to(dev).contiguous(), want_grad_obs=True)
    torch.cuda.synchronize()
    assert rel_l2(gobs.cpu(), grads[0]) < TOL, rel_l2(gobs.cpu(), grads[0])
    got = [t for pair in zip(gw, gb) for t in pair]
    for a, b in zip(got, grads[1:]):
        assert rel_l2(a.cpu(), b) < TOL


def test_spil_class_matches_reference(monkeypatch):
    """SPIL through create_alg on pyth_mobilerobot - what example_train/spil/spil_mlp_mobilerobot_offserial.py builds - one
    full update against the reference's, the model's noise source replaced by the reference run's recorded draws."""
    from test_alg_gpu import _load_alg
    alg, g, cfg = _load_alg("spil_mobilerobot")
    alg.gamma, alg.forward_step = cfg["gamma"], cfg["horizon"]
    alg.delta_i, alg.safe_prob_pre = np.array(g["state/delta_i"]), np.array(g["state/safe_prob_pre"])
    queue = [torch.from_numpy(g["in/noise_pev"]).cuda(), torch.from_numpy(g["in/noise_pim"]).cuda()]
    monkeypatch.setattr(hb, "mobilerobot_noise", lambda shape, device: queue.pop(0))
    data = data_from_golden(g)
    tb, info = alg.get_remote_update_info(data, 0)
    assert not queue
    assert abs(float(tb["Loss/Critic loss-RL iter"]) - float(g["pev_loss"])) <= TOL * max(1.0, abs(float(g["pev_loss"])))
    np.testing.assert_allclose(alg.safe_prob, g["safe_prob"], atol=1e-6)
    np.testing.assert_allclose(alg.lam, g["lam"], rtol=1e-5, atol=1e-6)
    assert abs(float(tb["Loss/Actor loss-RL iter"]) - float(g["pim_loss"])) <= TOL * max(1.0, abs(float(g["pim_loss"])))
    for i, gr in enumerate(info["v"]):
        assert rel_l2(gr.cpu(), g[f"pev_grad/{i}"]) < TOL, ("pev", i)
    for i, gr in enumerate(info["policy"]):
        assert rel_l2(gr.cpu(), g[f"pim_grad/{i}"]) < TOL, ("pim", i, rel_l2(gr.cpu(), g[f"pim_grad/{i}"]))
    # and with the model's own draws: finite, safe probability in [0, 1]
    monkeypatch.undo()
    tb, info = alg.get_remote_update_info(data, 1)
    assert all(torch.isfinite(gr).all() for gr in info["policy"]) and 0.0 <= float(alg.safe_prob[0]) <= 1.0


def test_wrapped_model_forward_draws_noise_on_the_device():
    """create_env_model("pyth_mobilerobot").forward: the obstacle moves with fresh draws every call; the ego part and the
    reward do not depend on them."""
    from gops_amd.create_pkg.create_env_model import create_env_model
    from gops_amd.utils.synthetic import make_batch
    model = create_env_model("pyth_mobilerobot", use_gpu=True)
    data = make_batch(dict(env_id="pyth_mobilerobot", batch=64), seed=2)
    obs, done = data["obs"].cuda(), data["done"].cuda()
    act = torch.rand(64, 2, device="cuda") * 2 - 1
    o1, r1, d1, i1 = model.forward(obs, act, done, {})
    o2, r2, d2, i2 = model.forward(obs, act, done, {})
    assert torch.equal(o1[:, :8], o2[:, :8]) and torch.equal(r1, r2)
    assert not torch.equal(o1[:, 8:], o2[:, 8:])
    assert i1["constraint"].shape == (64, 1)
    o3, _, _, _ = model.forward(obs, act, done, dict(noise=torch.zeros(64, 2, device="cuda")))
    ref, _, _, _ = orc.env_forward(orc.make_env("pyth_mobilerobot"), data["obs"], act.cpu(), data["done"], dict(noise=torch.zeros(1, 64, 2)))
    np.testing.assert_allclose(o3.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


def test_graph_replay_draws_fresh_noise(monkeypatch):
    """FHADPExterior on pyth_mobilerobot with the update captured into a HIP graph: the capture succeeds (the noise draw makes
    no host-to-device copy) and every replay moves the obstacle with new draws (the loss of the same batch changes)."""
    import warnings
    from gops_amd.create_pkg.create_alg import create_alg
    from gops_amd.utils.synthetic import make_batch
    from test_alg_gpu import _kwargs
    monkeypatch.setenv("GOPS_HIP_GRAPH", "1")
    cfg = dict(alg="FHADPExterior", env_id="pyth_mobilerobot", batch=64, horizon=8, hidden=(64, 64), act="elu", gamma=0.98)
    kw = _kwargs(cfg, dict(penalty=3.0), 5)
    kw.update(policy_func_name="FiniteHorizonPolicy", pre_horizon=8, policy_learning_rate=0.0)   # frozen weights: only the noise changes
    alg = create_alg(**kw)
    alg.networks.cuda()
    data = make_batch(cfg, 3)
    losses = []
    with warnings.catch_warnings():
        warnings.simplefilter("error")            # "HIP graph capture failed" would be a warning
        for it in range(6):
            tb = alg.local_update(data, it)
            losses.append(float(tb["Loss/Actor loss-RL iter"]))
    assert alg._update_graph.graph is not None
    assert all(np.isfinite(losses)) and len(set(losses[3:])) == 3, losses
