"""GPU: the algorithm classes built through the create_pkg factories against the reference
fixtures (FHADP / INFADP update API, state_dict layout, per-step env_model.forward)."""
import numpy as np
import pytest
import torch

from conftest import golden_meta, load_golden, rel_l2
from helpers import data_from_golden

from gops_amd import hip_backend as hb
from gops_amd.create_pkg.create_alg import create_alg
from gops_amd.utils.synthetic import act_dim_of, make_batch, obs_dim_of
from gops_amd.utils.tensorboard_setup import tb_tags

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _kwargs(cfg, extra, seed):
    A = act_dim_of(cfg)
    kw = dict(algorithm=cfg["alg"], trainer="off_serial_trainer", seed=seed, cnn_shared=False,
              env_id=cfg["env_id"], obsv_dim=obs_dim_of(cfg), action_dim=A, action_type="continu",
              action_high_limit=np.ones(A, dtype=np.float32), action_low_limit=-np.ones(A, dtype=np.float32),
              policy_func_type="MLP",
              policy_func_name="FiniteHorizonPolicy" if cfg["alg"] == "FHADP" else "DetermPolicy",
              policy_hidden_sizes=list(cfg["hidden"]), policy_hidden_activation=cfg["act"],
              policy_act_distribution="default", policy_learning_rate=1e-3, use_gpu=True)
    if cfg["alg"] in ("INFADP", "MAC", "SPIL"):
        kw.update(value_func_type="MLP", value_func_name="StateValue", value_hidden_sizes=list(cfg["hidden"]),
                  value_hidden_activation=cfg["act"], value_learning_rate=1e-3)
    if "pre_horizon" in cfg or cfg["alg"] == "FHADP":
        kw["pre_horizon"] = cfg.get("pre_horizon", cfg["horizon"])
    if "lq_config" in cfg:
        kw["lq_config"] = cfg["lq_config"]
    if cfg["alg"] == "SPIL":
        kw.update(policy_func_name="DetermPolicy")
        if "pre_horizon" in cfg:
            kw["pre_horizon"] = cfg["pre_horizon"]
    if cfg["alg"] == "MPG":
        kw.update(value_func_type="MLP", value_func_name="ActionValue", value_hidden_sizes=list(cfg["hidden"]),
                  value_hidden_activation=cfg["act"], value_learning_rate=1e-3)
    kw.update(extra)
    return kw


def _load_alg(name):
    g = load_golden(name)
    meta = golden_meta(g)
    cfg = meta["cfg"]
    alg = create_alg(**_kwargs(cfg, meta["extra"], meta["seed"]))
    sd = {k[3:]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith("sd/")}
    alg.load_state_dict(sd)      # the reference's checkpoint layout loads unchanged
    alg.networks.cuda()
    return alg, g, cfg


@pytest.mark.parametrize("name", ["fhadp_idp_gelu", "fhadp_veh_p10_elu", "fhadp_lq_s4a2_tanh",
                                  "fhadp_idp_selu_shaped", "fhadp_surrpen_p10_elu", "fhadp_lq_s3a1_obsscale",
                                  "fhadp_idp_obsscale_shift", "fhadp_veh2dof_p10_elu", "fhadp_veh_p10_refpara",
                                  "fhadp_idp_repeat2_gelu", "fhadp_pendulum_repeat3_tanh", "fhadp_veh_p10_nomask_elu"])
def test_fhadp_class_matches_reference(name):
    alg, g, cfg = _load_alg(name)
    alg.gamma = cfg["gamma"]
    data = data_from_golden(g)            # CPU batch, like a replay-buffer sample
    tb, info = alg.get_remote_update_info(data, 0)
    assert abs(float(tb["Loss/Actor loss-RL iter"]) - float(g["loss"])) <= TOL * max(1.0, abs(float(g["loss"])))
    assert "Time/Algorithm time [ms]-RL iter" in tb
    for i, gr in enumerate(info["grad"]):
        assert rel_l2(gr.cpu(), g[f"grad/{i}"]) < TOL
    # Adam step through the public API must move the weights and keep them finite
    before = [p.detach().clone() for p in alg.networks.policy.parameters()]
    alg.remote_update(info)
    after = list(alg.networks.policy.parameters())
    assert all(torch.isfinite(a).all() for a in after)
    assert any((a - b).abs().max() > 0 for a, b in zip(after, before))


@pytest.mark.parametrize("name", ["infadp_lq_s4a2_gelu", "infadp_veh_p10_relu", "infadp_lq_s5a1_obsscale_shift", "mac_idp_elu",
                                  "infadp_cartpole_gelu", "mac_pendulum_elu", "infadp_lq_s4a2_repeat3_elu",
                                  "infadp_cartpole_repeat2_relu", "infadp_cartpole_nomask_relu", "infadp_veh2dof_nomask_gelu"])
def test_infadp_class_matches_reference(name):
    alg, g, cfg = _load_alg(name)
    alg.gamma, alg.forward_step = cfg["gamma"], cfg["horizon"]
    data = data_from_golden(g)
    tb, info = alg.get_remote_update_info(data, 0)       # PEV
    assert list(info) == ["v"]
    assert abs(float(tb["Loss/Critic loss-RL iter"]) - float(g["pev_loss"])) <= TOL * max(1.0, abs(float(g["pev_loss"])))
    for i, gr in enumerate(info["v"]):
        assert rel_l2(gr.cpu(), g[f"pev_grad/{i}"]) < TOL
    tb, info = alg.get_remote_update_info(data, 1)       # PIM
    assert list(info) == ["policy"]
    assert abs(float(tb["Loss/Actor loss-RL iter"]) - float(g["pim_loss"])) <= TOL * max(1.0, abs(float(g["pim_loss"])))
    for i, gr in enumerate(info["policy"]):
        assert rel_l2(gr.cpu(), g[f"pim_grad/{i}"]) < TOL
    # local_update applies Adam + Polyak to the updated net only
    vt_before = [p.detach().clone() for p in alg.networks.v_target.parameters()]
    pt_before = [p.detach().clone() for p in alg.networks.policy_target.parameters()]
    alg.local_update(data, 0)
    assert any((a - b).abs().max() > 0 for a, b in zip(alg.networks.v_target.parameters(), vt_before))
    assert all((a - b).abs().max() == 0 for a, b in zip(alg.networks.policy_target.parameters(), pt_before))


def test_env_model_forward_contract():
    from gops_amd.create_pkg.create_env_model import create_env_model
    g = load_golden("step_veh_p10")
    model = create_env_model("pyth_veh3dofconti", pre_horizon=10, use_gpu=True)
    data = {k: v.cuda() for k, v in data_from_golden(g).items()}
    info = {k: data[k] for k in ("state", "ref_points", "path_num", "u_num", "ref_time")}
    obs, r, d, ninfo = model.forward(data["obs"], torch.from_numpy(g["s0/act"]).cuda(), data["done"], info)
    assert d.dtype == torch.bool and set(ninfo) >= {"state", "ref_points", "ref_time", "path_num", "u_num"}
    assert rel_l2(obs.cpu(), g["s0/obs"]) < TOL and rel_l2(r.cpu(), g["s0/rew"]) < TOL
    with pytest.raises(RuntimeError):
        model.forward(data["obs"].cpu(), torch.from_numpy(g["s0/act"]), data["done"].cpu(), {})


def test_trainer_loop_runs_and_learns(tmp_path):
    """on_serial_trainer end to end on the GPU: the LQ return improves over a few dozen updates."""
    from gops_amd.create_pkg.create_trainer import create_trainer
    from gops_amd.trainer.sampler.initial_state_sampler import InitialStateSampler
    cfg = dict(alg="FHADP", env_id="pyth_lq", lq_config="s4a2", batch=256, horizon=20, hidden=(64, 64),
               act="gelu", gamma=1.0)
    torch.manual_seed(0)
    kw = _kwargs(cfg, {}, 0)
    kw.update(trainer="on_serial_trainer", max_iteration=60, log_save_interval=1000, apprfunc_save_interval=1000,
              eval_interval=10 ** 9, save_folder=str(tmp_path), ini_network_dir=None)
    alg = create_alg(**kw)
    trainer = create_trainer(alg, InitialStateSampler(cfg, seed=5, device="cuda"), None, None, **kw)
    losses = []
    for _ in range(60):
        trainer.step()
        trainer.iteration += 1
        losses.append(alg.tb_info["Loss/Actor loss-RL iter"])
    assert np.mean(losses[-10:]) < np.mean(losses[:10])
    trainer.save_apprfunc()
    assert (tmp_path / "apprfunc" / "apprfunc_60.pkl").exists()


@pytest.mark.gpu
def test_hip_adam_matches_torch_adam():
    """gops_adam_step vs torch.optim.Adam (the reference's optimizer, fhadp.py:45-47) over 25 steps."""
    from gops_amd.hip_backend import HipAdam
    g = torch.Generator().manual_seed(3)
    shapes = [(256, 47), (256,), (256, 256), (256,), (2, 256), (2,)]
    p0 = [torch.randn(s, generator=g) * 0.1 for s in shapes]
    pa = [torch.nn.Parameter(p.clone().cuda()) for p in p0]
    pb = [torch.nn.Parameter(p.clone().cuda()) for p in p0]
    oa, ob = HipAdam(pa, lr=3e-4), torch.optim.Adam(pb, lr=3e-4)
    for it in range(25):
        for a, b in zip(pa, pb):
            gr = (torch.randn(a.shape, generator=g) * (10.0 ** (it % 5 - 3))).cuda()
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    torch.cuda.synchronize()
    for a, b in zip(pa, pb):
        assert rel_l2(a.detach().cpu(), b.detach().cpu()) < 1e-6
        assert torch.allclose(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"], rtol=1e-5, atol=0)
    # optimizer checkpoints are interchangeable with torch's Adam
    ob2 = torch.optim.Adam(pb, lr=3e-4)
    ob2.load_state_dict(oa.state_dict())
    assert int(ob2.state[pb[0]]["step"]) == 25


@pytest.mark.gpu
def test_graph_replay_matches_eager_updates(monkeypatch):
    """The captured HIP graph of (gradient + Adam) replays exactly the eager update: two FHADP
    learners, one with capture disabled, stay bit-identical over fresh batches and an lr change."""
    from gops_amd.utils.synthetic import CONFIGS, make_batch
    from bench import alg_kwargs
    cfg = dict(CONFIGS["target_veh3dof_fhadp_b4096_h30"], batch=96, horizon=12, pre_horizon=12)
    algs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("GOPS_HIP_GRAPH", flag)
        torch.manual_seed(5)
        alg = create_alg(**alg_kwargs(cfg, 0))
        alg.networks.to("cuda")
        algs.append(alg)
    for it in range(8):
        data = {k: v.cuda() for k, v in make_batch(cfg, 50 + it).items()}
        if it == 5:
            for alg in algs:
                alg.networks.policy_optimizer.param_groups[0]["lr"] *= 0.5
        infos = []
        for alg, flag in zip(algs, ("1", "0")):
            monkeypatch.setenv("GOPS_HIP_GRAPH", flag)
            infos.append(dict(alg.local_update(data, it)))
        assert infos[0]["Loss/Actor loss-RL iter"] == infos[1]["Loss/Actor loss-RL iter"]
    assert algs[0]._update_graph.graph is not None and algs[1]._update_graph.graph is None
    for a, b in zip(algs[0].networks.policy.parameters(), algs[1].networks.policy.parameters()):
        assert torch.equal(a, b)
    st = algs[0].networks.policy_optimizer.state_dict()["state"]
    assert all(int(v["step"]) == 8 for v in st.values())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["fhadp2_idp_gelu", "fhadp2_veh_p10_elu"])
def test_fhadp2_class_matches_reference(name):
    """gops_amd FHADP2 (create_alg surface) loaded with the reference's weights reproduces its loss and
    per-parameter gradients, then takes an Adam step."""
    g = load_golden(name)
    meta = golden_meta(g)
    cfg = meta["cfg"]
    kw = _kwargs(cfg, meta["extra"], meta["seed"])
    kw.update(algorithm="FHADP2", policy_func_name="FiniteHorizonFullPolicy")
    alg = create_alg(**kw)
    alg.gamma = cfg["gamma"]
    sd = {k[3:]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith("sd/")}
    alg.load_state_dict(sd)
    alg.networks.to("cuda")
    data = data_from_golden(g)
    tb, info = alg.get_remote_update_info(data, 0)
    assert abs(float(tb["Loss/Actor loss-RL iter"]) - float(g["loss"])) <= 1e-4 * max(1.0, abs(float(g["loss"])))
    for i, gr in enumerate(info["grad"]):
        assert rel_l2(gr.cpu(), g[f"grad/{i}"]) < 1e-4, (name, i)
    before = [p.detach().clone() for p in alg.networks.policy.parameters()]
    alg.local_update(data, 1)
    assert any(not torch.equal(a, b) for a, b in zip(before, alg.networks.policy.parameters()))


@pytest.mark.gpu
def test_off_serial_trainer_with_device_replay_buffer(tmp_path):
    """off_serial_trainer + HBM-resident replay buffer + FHADP on the GPU: replay batches are sampled
    on the device (never on the host), a numpy-side sampler gets a host copy of the weights, and the
    veh3dof tracking return improves."""
    from gops_amd.create_pkg.create_buffer import create_buffer
    from gops_amd.create_pkg.create_trainer import create_trainer
    from gops_amd.utils.synthetic import make_batch
    cfg = dict(alg="FHADP", env_id="pyth_veh3dofconti", batch=2048, horizon=10, pre_horizon=10, hidden=(64, 64),
               act="elu", gamma=1.0)
    torch.manual_seed(0)
    kw = _kwargs(cfg, {}, 0)
    info_spec = {"state": {"shape": (6,), "dtype": np.float32}, "ref_points": {"shape": (11, 4), "dtype": np.float32},
                 "path_num": {"shape": (), "dtype": np.float32}, "u_num": {"shape": (), "dtype": np.float32},
                 "ref_time": {"shape": (), "dtype": np.float32}}
    kw.update(trainer="off_serial_trainer", buffer_name="replay_buffer", buffer_max_size=4096, buffer_warm_size=2048,
              replay_batch_size=256, sample_interval=4, additional_info=info_spec, max_iteration=50,
              log_save_interval=1000, apprfunc_save_interval=1000, eval_interval=10 ** 9, save_folder=str(tmp_path),
              ini_network_dir=None)
    alg = create_alg(**kw)   # networks start on the CPU, as create_alg leaves them: the trainer moves them once
    buf = create_buffer(**kw)
    assert buf.device.type == "cuda"

    class HostSampler:   # stands in for the reference's numpy-env sampler: uses `networks` on the CPU
        networks = None
        calls = 0

        def sample(self):
            HostSampler.calls += 1
            data = make_batch(dict(cfg, batch=64), 300 + HostSampler.calls)
            with torch.no_grad():
                act = self.networks.policy(data["obs"], 1)          # CPU forward of the host copy
            assert act.device.type == "cpu"
            out = []
            for i in range(64):
                info = {k: data[k][i].numpy() for k in info_spec}
                out.append((data["obs"][i].numpy(), act[i].numpy(), 0.0, False, info, data["obs"][i].numpy(), info, 0.0))
            return out, {"Time/Sampler time [ms]-RL iter": 0.0}

        def get_total_sample_number(self):
            return 64 * HostSampler.calls

    trainer = create_trainer(alg, HostSampler(), buf, None, **kw)
    assert len(buf) >= 2048 and next(alg.networks.parameters()).is_cuda
    p0 = next(alg.networks.policy.parameters()).data_ptr()
    losses = []
    for _ in range(50):
        trainer.step()
        trainer.iteration += 1
        losses.append(alg.tb_info["Loss/Actor loss-RL iter"])
    assert next(alg.networks.policy.parameters()).data_ptr() == p0 and next(alg.networks.parameters()).is_cuda
    batch = buf.sample_batch(8)
    assert all(v.is_cuda and v.dtype == torch.float32 for v in batch.values())
    assert np.mean(losses[-10:]) < np.mean(losses[:10])
    # the sampler's host copy follows the learner (deep copy taken AFTER the HIP path cached its C-ABI views of the
    # learner's modules - those views live outside the modules and are not copied)
    host_w = next(trainer._host_networks.policy.parameters())
    assert not host_w.is_cuda
    assert torch.equal(host_w, next(alg.networks.policy.parameters()).cpu()) or trainer.iteration % kw["sample_interval"] != 1


@pytest.mark.gpu
@pytest.mark.parametrize("env_id", ["pyth_idpendulum", "pyth_veh3dofconti"])
def test_device_env_sampler_closed_loop(env_id, tmp_path):
    """N environments stepped together on the GPU by gops_env_step: every stored transition equals the
    oracle's wrapped model step of (obs, act), episodes restart on termination / time-out, and the
    batches flow into the device replay buffer through the off-policy trainer."""
    from helpers import oracle_env
    from oracle import adp_oracle as orc
    from gops_amd.create_pkg.create_buffer import create_buffer
    from gops_amd.create_pkg.create_trainer import create_trainer
    from gops_amd.trainer.sampler.device_env_sampler import DeviceEnvSampler
    cfg = dict(alg="FHADP", env_id=env_id, batch=128, horizon=10, pre_horizon=10, hidden=(64, 64), act="gelu", gamma=1.0)
    torch.manual_seed(1)
    kw = _kwargs(cfg, {}, 1)
    alg = create_alg(**kw)
    alg.networks.to("cuda")
    smp = DeviceEnvSampler(cfg, alg.envmodel, n_envs=96, steps_per_sample=6, max_episode_steps=4, seed=11)
    smp.networks = alg.networks
    batch, _ = smp.sample()
    n = 96 * 6
    assert all(v.is_cuda and v.shape[0] == n for v in batch.values())
    env = oracle_env(cfg, {})
    info = {k: batch[k].cpu() for k in ("state", "ref_points", "path_num", "u_num", "ref_time") if k in batch}
    o2, r, d, ninfo = orc.env_forward(env, batch["obs"].cpu(), batch["act"].cpu(), torch.zeros(n), info)
    assert rel_l2(batch["obs2"].cpu(), o2) < 1e-4 and rel_l2(batch["rew"].cpu(), r) < 1e-4
    assert np.array_equal(batch["done"].cpu().numpy() != 0, d.numpy())
    if env_id == "pyth_veh3dofconti":
        assert rel_l2(batch["next_state"].cpu(), ninfo["state"]) < 1e-4
        assert rel_l2(batch["next_ref_time"].cpu(), ninfo["ref_time"]) < 1e-6
    # time-outs after 4 steps: step 4 of every instance starts from a fresh reset state, not from obs2 of step 3
    obs_s = batch["obs"].view(6, 96, -1)
    obs2_s = batch["obs2"].view(6, 96, -1)
    assert torch.equal(obs_s[1], obs2_s[0]) or (batch["done"].view(6, 96)[0] != 0).any()
    assert not torch.equal(obs_s[4], obs2_s[3])
    assert smp.get_total_sample_number() == n
    # into the replay buffer through the trainer
    info_spec = {k: {"shape": tuple(v.shape[1:]), "dtype": np.float32} for k, v in info.items()}
    kw.update(trainer="off_serial_trainer", buffer_name="replay_buffer", buffer_max_size=4096, buffer_warm_size=1000,
              replay_batch_size=128, sample_interval=1, additional_info=info_spec, max_iteration=5,
              log_save_interval=1000, apprfunc_save_interval=1000, eval_interval=10 ** 9, save_folder=str(tmp_path),
              ini_network_dir=None)
    buf = create_buffer(**kw)
    trainer = create_trainer(alg, smp, buf, None, **kw)
    assert len(buf) >= 1000
    for _ in range(5):
        trainer.step()
        trainer.iteration += 1
    assert len(buf) >= 1000 + 5 * n and np.isfinite(alg.tb_info["Loss/Actor loss-RL iter"])


@pytest.mark.gpu
def test_device_env_sampler_steps_the_mobilerobot_data_env(monkeypatch):
    """pyth_mobilerobot under DeviceEnvSampler(env_step="data"): resets come from the data env's own reset box, every stored
    transition is the oracle's DATA-env step of (obs, act) with the obstacle draws the step used, headings stay within +-pi, and
    terminated instances restart."""
    from helpers import oracle_env
    from oracle import adp_oracle as orc
    from gops_amd import hip_backend as hb
    from gops_amd.trainer.sampler.device_env_sampler import DeviceEnvSampler
    cfg = dict(alg="FHADP", env_id="pyth_mobilerobot", batch=64, horizon=4, hidden=(64, 64), act="relu", gamma=0.99)
    torch.manual_seed(2)
    alg = create_alg(**_kwargs(cfg, {}, 2))
    alg.networks.to("cuda")
    draws = []
    real = hb.mobilerobot_noise
    monkeypatch.setattr(hb, "mobilerobot_noise", lambda shape, device: draws.append(real(shape, device)) or draws[-1])
    smp = DeviceEnvSampler(cfg, alg.envmodel, n_envs=256, steps_per_sample=40, max_episode_steps=30, seed=5, noise_std=0.6)
    first = smp.obs.cpu()
    assert (first[:, 0] >= 0).all() and (first[:, 0] <= 2.7).all() and (first[:, 4] == 0).all() and (first[:, 8] >= 3.5).all()
    assert torch.equal(first[:, 5:8], torch.stack((first[:, 1], first[:, 2], first[:, 3] - 0.3), 1))
    smp.networks = alg.networks
    batch, _ = smp.sample()
    n = 256 * 40
    assert len(draws) == 40 and set(batch) == {"obs", "act", "rew", "done", "obs2", "logp"}
    env = oracle_env(cfg, {})
    o2, r, d, _ = orc.data_env_forward(env, batch["obs"].cpu(), batch["act"].cpu(), dict(noise=torch.cat(draws).cpu()))
    assert rel_l2(batch["obs2"].cpu(), o2) < 1e-5 and rel_l2(batch["rew"].cpu(), r) < 1e-5
    assert np.array_equal(batch["done"].cpu().numpy() != 0, d.numpy() != 0)
    assert batch["obs2"][:, [2, 10]].abs().max().item() <= np.float32(np.pi)
    # restarts: a finished / timed-out instance continues from a fresh reset state
    obs_s, obs2_s, done_s = batch["obs"].view(40, 256, -1), batch["obs2"].view(40, 256, -1), batch["done"].view(40, 256)
    cont = (done_s[:29] == 0)
    assert torch.equal(obs_s[1:30][cont], obs2_s[:29][cont])
    never = (done_s[:30] == 0).all(0)   # these ran into the 30-step time-out together
    assert never.sum() > 100 and (obs_s[30][never][:, 4] == 0).all() and (obs2_s[29][never][:, 4] != 0).any()


@pytest.mark.gpu
def test_infadp_graph_replay_matches_eager_updates(monkeypatch):
    """INFADP's PEV and PIM updates (gradient + Adam + Polyak) captured as HIP graphs replay exactly
    the eager updates: two learners stay bit-identical over alternating iterations and fresh batches."""
    from gops_amd.utils.synthetic import make_batch
    cfg = dict(alg="INFADP", env_id="pyth_idpendulum", batch=96, horizon=8, hidden=(64, 64), act="gelu", gamma=0.99)
    algs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("GOPS_HIP_GRAPH", flag)
        torch.manual_seed(9)
        alg = create_alg(**_kwargs(cfg, {}, 9))
        alg.forward_step = cfg["horizon"]
        alg.networks.to("cuda")
        algs.append(alg)
    for it in range(12):
        data = {k: v.cuda() for k, v in make_batch(cfg, 70 + it).items()}
        infos = []
        for alg, flag in zip(algs, ("1", "0")):
            monkeypatch.setenv("GOPS_HIP_GRAPH", flag)
            infos.append(dict(alg.local_update(data, it)))
        strip = lambda d: {k: v for k, v in d.items() if "time" not in k.lower()}
        assert strip(infos[0]) == strip(infos[1])
    assert all(c.graph is not None for c in algs[0]._graphs.values()) and len(algs[0]._graphs) == 2
    assert all(c.graph is None for c in algs[1]._graphs.values())
    for a, b in zip(algs[0].networks.parameters(), algs[1].networks.parameters()):
        assert torch.equal(a, b)


CSTR_ALG_CASES = ["fhadp_ext_surrcstr", "fhadp_int_surrcstr", "fhadp_lag_surrcstr", "fhadp_int_detour", "fhadp_ext_detour",
                  "fhadp_ext_surrpen", "fhadp_int_surrpen", "fhadp_ext_errcstr", "fhadp_lag_errcstr",
                  "fhadp_int_veh2dof_errcstr"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", CSTR_ALG_CASES)
def test_constrained_fhadp_classes_match_reference(name):
    """FHADPExterior / FHADPInterior / FHADPLagrangian (create_alg surface) on the constrained veh3dofconti models, loaded
    with the reference's weights: total / reward / constraint losses, feasible ratio and every policy gradient at 1e-4
    against the reference's `_compute_loss_policy` + backward; then the penalty / multiplier schedules advance."""
    g = load_golden(name)
    meta = golden_meta(g)
    cfg, extra = meta["cfg"], meta["extra"]
    kw = _kwargs(cfg, {}, meta["seed"])
    kw.update(extra)
    kw.update(algorithm=cfg["alg"], policy_func_name="FiniteHorizonPolicy", pre_horizon=cfg["pre_horizon"])
    if "surr_veh_num" in cfg:
        kw["surr_veh_num"] = cfg["surr_veh_num"]
    alg = create_alg(**kw)
    alg.gamma = cfg["gamma"]
    sd = {k[3:]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith("sd/")}
    alg.load_state_dict(sd)
    alg.networks.to("cuda")
    data = data_from_golden(g)
    tb, info = alg.get_remote_update_info(data, 0)
    for key in ("Loss/Actor loss-RL iter", "Loss/Actor reward loss-RL iter", "Loss/Actor constraint loss-RL iter",
                "Loss/Feasible ratio-RL iter", "Loss/Penalty coefficient-RL iter", "Loss/Lagrange multiplier-RL iter"):
        if "tb/" + key in g:
            want = float(g["tb/" + key])
            assert abs(float(tb[key]) - want) <= 1e-4 * max(1.0, abs(want)), (name, key, float(tb[key]), want)
    for i, gr in enumerate(info["grad"]):
        assert rel_l2(gr.cpu(), g[f"grad/{i}"]) < 1e-4, (name, i, rel_l2(gr.cpu(), g[f"grad/{i}"]))
    # schedules: penalty grows every `penalty_delay` updates / the multiplier takes an ascent step every `multiplier_delay`
    before = [p.detach().clone() for p in alg.networks.policy.parameters()]
    if hasattr(alg, "penalty"):
        alg.penalty_delay, p0 = 2, alg.penalty
        alg.local_update(data, 1)
        assert alg.penalty == pytest.approx(p0 * alg.penalty_increase)
    else:
        alg.multiplier_delay, m0 = 2, alg.multiplier
        alg.local_update(data, 1)
        assert alg.multiplier > m0          # violation > 0: the multiplier rises
    assert any(not torch.equal(a, b) for a, b in zip(before, alg.networks.policy.parameters()))
    assert all(torch.isfinite(p).all() for p in alg.networks.policy.parameters())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["fhadp_ext_surrcstr", "fhadp_int_detour", "fhadp_lag_surrcstr"])
def test_constrained_fhadp_schedule_advances_under_graph_replay(name, monkeypatch):
    """The penalty / multiplier schedule is host-side bookkeeping of every computed gradient (reference:
    fhadp_exterior.py:68-70, fhadp_interior.py:80-82, fhadp_lagrangian.py:72-77).  With the update captured as a HIP
    graph the coefficient is a graph INPUT (it travels with the batch), so over 9 updates a graph learner and an eager
    learner walk the same schedule and stay bit-identical, and the graph is captured once and kept."""
    g = load_golden(name)
    meta = golden_meta(g)
    cfg, extra = meta["cfg"], meta["extra"]
    sd = {k[3:]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith("sd/")}
    data = data_from_golden(g)
    algs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("GOPS_HIP_GRAPH", flag)
        kw = _kwargs(cfg, {}, meta["seed"])
        kw.update(extra)
        kw.update(algorithm=cfg["alg"], policy_func_name="FiniteHorizonPolicy", pre_horizon=cfg["pre_horizon"])
        if "surr_veh_num" in cfg:
            kw["surr_veh_num"] = cfg["surr_veh_num"]
        alg = create_alg(**kw)
        alg.gamma = cfg["gamma"]
        alg.load_state_dict(sd)
        alg.networks.to("cuda")
        if hasattr(alg, "penalty"):
            alg.penalty_delay = 2
        else:
            alg.multiplier_delay = 2
        algs.append(alg)
    coef = lambda a: a.penalty if hasattr(a, "penalty") else a.multiplier
    c0 = coef(algs[0])
    trace = []
    for it in range(9):
        infos = []
        for alg, flag in zip(algs, ("1", "0")):
            monkeypatch.setenv("GOPS_HIP_GRAPH", flag)
            infos.append(dict(alg.local_update(data, it)))
        strip = lambda d: {k: v for k, v in d.items() if "time" not in k.lower()}
        assert strip(infos[0]) == strip(infos[1]), (it, infos)
        assert coef(algs[0]) == coef(algs[1])
        trace.append(coef(algs[0]))
    assert algs[0].update_step == algs[1].update_step == 9
    assert algs[0]._update_graph.graph is not None and algs[1]._update_graph.graph is None
    if hasattr(algs[0], "penalty"):   # four schedule steps in nine updates (delay 2)
        assert trace[-1] == pytest.approx(c0 * algs[0].penalty_increase ** 4)
    else:                             # violation > 0 in this fixture: the multiplier keeps rising, also under replay
        assert trace[-1] > trace[4] > c0
    for a, b in zip(algs[0].networks.parameters(), algs[1].networks.parameters()):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(env_id="pyth_lq", lq_config="s4a2"), dict(env_id="pyth_lq", lq_config="s6a3"),
                                 dict(env_id="pyth_idpendulum"), dict(env_id="pyth_veh3dofconti", pre_horizon=10)],
                         ids=lambda c: c["env_id"][5:] + c.get("lq_config", ""))
def test_opt_controller_cost_and_jacobian_match_raw_model(cfg):
    """OptController: cost and Jacobian of a shooting rollout over the RAW model (no wrappers, no
    MaskAtDone) from ONE forward + ONE backward kernel launch, against the oracle's step-by-step rollout + autograd."""
    from gops_amd.create_pkg.create_env_model import create_env_model
    from gops_amd.sys_simulator.opt_controller import OptController
    from gops_amd.utils.synthetic import make_batch
    from oracle import adp_oracle as orc
    model = create_env_model(**cfg, use_gpu=True)
    T, interval = 12, 3
    ctrl = OptController(model, num_pred_step=T, ctrl_interval=interval, gamma=0.97, mode="shooting")
    data = make_batch(dict(cfg, batch=3), 21)
    env = orc.make_env(cfg["env_id"], lq_config=cfg.get("lq_config", "s4a2"), pre_horizon=cfg.get("pre_horizon", 10))
    rng = np.random.RandomState(4)
    lo, hi = ctrl.bounds.lb, ctrl.bounds.ub
    for b in range(3):
        u = rng.uniform(0.6 * lo, 0.6 * hi)                      # inside the action bounds, in model units
        info = {k: data[k][b].numpy() for k in ("state", "ref_points", "path_num", "u_num", "ref_time") if k in data}
        cost, jac = ctrl._cost_fcn_and_jac(u, data["obs"][b].numpy(), info)
        acts = torch.tensor(u, dtype=torch.float32).reshape(T // interval, -1).repeat_interleave(interval, 0)[None]
        oinfo = {k: data[k][b:b + 1] for k in info}
        want_cost, want_obs, want_jac = orc.raw_shooting_cost(env, data["obs"][b:b + 1], oinfo, acts, 0.97)
        assert abs(cost - want_cost.item()) <= 1e-4 * max(1.0, abs(want_cost.item())), (cfg, cost, want_cost)
        want = want_jac[0].reshape(T // interval, interval, -1).sum(1).reshape(-1)
        assert rel_l2(jac, want) < 1e-4, (cfg, rel_l2(jac, want))
        fin, stage = ctrl._rollout(u, data["obs"][b].numpy(), info)
        # veh3dofconti: the final observation holds 10 appended reference headings (fp32 finite differences, see
        # test_env_step_vs_reference_fixture) - ego part exact, whole vector to 1e-3
        assert rel_l2(fin.cpu()[:6], want_obs[0][:6]) < 1e-4 and rel_l2(fin.cpu(), want_obs[0]) < 1e-3
        assert abs(stage.sum().item() - want_cost.item()) <= 1e-4 * max(1.0, abs(want_cost.item()))


_CSTR_MODELS = [dict(env_id="pyth_veh3dofconti_surrcstr", pre_horizon=10), dict(env_id="pyth_veh3dofconti_detour", pre_horizon=10),
                dict(env_id="pyth_veh3dofconti_errcstr", pre_horizon=10), dict(env_id="pyth_veh2dofconti_errcstr", pre_horizon=10)]
_INFO_ALL = ("state", "ref_points", "path_num", "u_num", "ref_time", "surr_state")


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", _CSTR_MODELS, ids=lambda c: c["env_id"][5:])
def test_get_constraint_matches_oracle(cfg):
    """model.get_constraint(obs, info) (gops_env_constraint) of the constrained vehicle models against the oracle's
    restatement of the reference hooks."""
    from gops_amd.create_pkg.create_env_model import create_env_model
    from gops_amd.utils.synthetic import make_batch
    from oracle import adp_oracle as orc
    model = create_env_model(**cfg, use_gpu=True)
    assert model.get_constraint is not None
    data = make_batch(dict(cfg, batch=70), 12)
    env = orc.make_env(cfg["env_id"], pre_horizon=10)
    got = model.get_constraint(data["obs"].cuda(), {k: data[k].cuda() for k in ("state", "surr_state") if k in data})
    if env["kind"] == "veh_err":
        want = torch.stack((data["obs"][:, 1].abs() - 0.2, data["obs"][:, 3].abs() - 2.0), 1)
    elif env["kind"] == "veh2":
        want = data["obs"][:, 0:1].abs() - 0.2
    else:
        want = orc.surr_constraint(env, data["state"], data["surr_state"])
    assert got.shape == want.shape
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-5, atol=2e-5)
    with pytest.raises(RuntimeError, match="no CPU path"):
        model.get_constraint(data["obs"], {k: data[k] for k in ("state", "surr_state") if k in data})
    assert create_env_model("pyth_lq", lq_config="s4a2").get_constraint is None


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", _CSTR_MODELS, ids=lambda c: c["env_id"][5:])
def test_opt_controller_constraint_function_and_jacobian_match_raw_model(cfg):
    """OptController's inequality constraints (opt_controller.py:178-206): -get_constraint on all T + 1 states of a shooting
    rollout over the raw model and its Jacobian w.r.t. the held actions - one rollout launch (+ gops_env_constraint for the
    initial state) and one forward + backward launch over T n_c seeded replicas - against the oracle's step-by-step rollout
    with one autograd pass per row."""
    from gops_amd.create_pkg.create_env_model import create_env_model
    from gops_amd.sys_simulator.opt_controller import OptController
    from gops_amd.utils.synthetic import make_batch
    from oracle import adp_oracle as orc
    model = create_env_model(**cfg, use_gpu=True)
    T, interval = 8, 2
    ctrl = OptController(model, num_pred_step=T, ctrl_interval=interval, gamma=0.97, mode="shooting")
    data = make_batch(dict(cfg, batch=3), 33)
    env = orc.make_env(cfg["env_id"], pre_horizon=10)
    rng = np.random.RandomState(6)
    lo, hi = ctrl.bounds.lb, ctrl.bounds.ub
    nc, n, A = env["n_constraint"], T // interval, env["act_dim"]
    for b in range(3):
        u = rng.uniform(0.6 * lo, 0.6 * hi)
        info = {k: data[k][b].numpy() for k in _INFO_ALL if k in data}
        vec = ctrl._constraint_fcn(u, data["obs"][b].numpy(), info)
        jac = ctrl._constraint_jac(u, data["obs"][b].numpy(), info)
        acts = torch.tensor(u, dtype=torch.float32).reshape(n, -1).repeat_interleave(interval, 0)[None]
        want_vec, want_jac = orc.raw_shooting_constraints(env, data["obs"][b:b + 1], {k: data[k][b:b + 1] for k in info}, acts)
        assert vec.shape == ((T + 1) * nc,) and jac.shape == ((T + 1) * nc, n * A)
        np.testing.assert_allclose(vec, want_vec.numpy(), rtol=1e-4, atol=1e-4)
        want = want_jac.reshape((T + 1) * nc, n, interval, A).sum(2).reshape((T + 1) * nc, n * A).numpy()
        assert rel_l2(jac, want) < 1e-4, (cfg, b, rel_l2(jac, want))
        assert np.abs(jac[:nc]).max() == 0.0   # the initial state does not depend on the actions
        # the cost side of the same controller still matches the raw model (surrounding-vehicle kernels in open-loop mode)
        if env["kind"] in ("veh_err", "veh2"):
            cost, cjac = ctrl._cost_fcn_and_jac(u, data["obs"][b].numpy(), info)
            assert np.isfinite(cost) and np.all(np.isfinite(cjac))


@pytest.mark.gpu
def test_opt_controller_keeps_the_tracking_error_constraint():
    """Receding-horizon solve on pyth_veh2dofconti_errcstr from a state that drifts away from the reference path: the tube
    |delta_y| <= tol is chosen between the lateral error the state cannot avoid (step 1) and the largest one of the
    UNCONSTRAINED optimum; the constrained solve (SLSQP on kernel constraint values / Jacobians) then keeps every predicted
    state inside the tube at a higher cost."""
    import scipy.optimize as opt
    from gops_amd.create_pkg.create_env_model import create_env_model
    from gops_amd.sys_simulator.opt_controller import OptController
    from gops_amd.utils.synthetic import make_batch
    cfg = dict(env_id="pyth_veh2dofconti_errcstr", pre_horizon=10)
    data = make_batch(dict(cfg, batch=1), 1)
    state, ref = data["state"][0].clone(), data["ref_points"][0]
    state[0], state[1], state[2], state[3] = ref[0, 0] + 0.05, ref[0, 1] + 0.2, 0.2, 0.0   # heading 0.2 rad off the path
    obs = torch.cat((state[:2] - ref[0], state[2:], state[:1] - ref[1:, 0])).numpy()
    info = {k: data[k][0].numpy() for k in ("ref_points", "path_num", "u_num", "ref_time")}
    info["state"] = state.numpy()
    T = 10
    mk = lambda tol: OptController(create_env_model(**cfg, y_error_tol=tol, use_gpu=True), num_pred_step=T, ctrl_interval=1, gamma=1.0,
                                   mode="shooting", minimize_options={"maxiter": 200, "ftol": 1e-9})
    probe = mk(0.2)
    free = opt.minimize(probe._cost_fcn_and_jac, np.zeros(T), args=(obs, info), jac=True, bounds=probe.bounds, method="L-BFGS-B")
    err_free = 0.2 - probe._constraint_fcn(free.x, obs, info)        # |delta_y| of the T + 1 predicted states
    unavoidable, worst = float(err_free[:2].max()), float(err_free[2:].max())
    assert worst > unavoidable + 0.004, ("the scenario no longer drifts outwards", err_free)
    tol = unavoidable + 0.7 * (worst - unavoidable)
    ctrl = mk(tol)
    ctrl(obs, info)
    sol = ctrl.last_result
    c_sol = -ctrl._constraint_fcn(sol.x, obs, info)
    assert sol.success and c_sol.max() <= 1e-5, (sol.message, c_sol)
    assert np.all(sol.x >= ctrl.bounds.lb - 1e-9) and np.all(sol.x <= ctrl.bounds.ub + 1e-9)
    assert (-ctrl._constraint_fcn(free.x, obs, info)).max() > 1e-3      # the unconstrained optimum leaves this tube ...
    assert free.fun < sol.fun                                             # ... and staying inside costs something


@pytest.mark.gpu
def test_opt_controller_solves_lq_regulation():
    """Receding-horizon control of pyth_lq s4a2 from a perturbed state: every solve lowers its cost below the zero-input
    cost, respects the action bounds, and the closed loop (model steps on the GPU) regulates the state."""
    from gops_amd.create_pkg.create_env_model import create_env_model
    from gops_amd.sys_simulator.opt_controller import OptController
    from gops_amd import hip_backend as hb
    model = create_env_model("pyth_lq", lq_config="s4a2", use_gpu=True)
    ctrl = OptController(model, num_pred_step=20, ctrl_interval=2, gamma=1.0, minimize_options={"maxiter": 60}, mode="shooting")
    x = np.array([1.5, -0.6, 1.0, 0.5], dtype=np.float32)
    zero_cost, _ = ctrl._cost_fcn_and_jac(np.zeros_like(ctrl.initial_guess), x, {})
    raw = ctrl._rollout_obj.desc.env
    norms = [float(np.linalg.norm(x))]
    for _ in range(25):
        u = ctrl(x)
        assert np.all(u >= ctrl.bounds.lb[:2] - 1e-9) and np.all(u <= ctrl.bounds.ub[:2] + 1e-9)
        assert ctrl.last_result.fun < zero_cost or norms[-1] < 0.2
        # one step of the raw model with the (model-unit) action: min/max_action of `raw` are the identity scaling
        o = torch.tensor(x, device="cuda").reshape(1, -1)
        nobs, _, _, _ = hb.env_step(raw, o, torch.tensor(u, dtype=torch.float32, device="cuda").reshape(1, -1),
                                    torch.zeros(1, device="cuda"), {})
        x = nobs[0].cpu().numpy()
        norms.append(float(np.linalg.norm(x)))
    assert norms[-1] < 0.8 * norms[0] and all(b < a + 1e-6 for a, b in zip(norms, norms[1:])), norms   # 2.5 s of a slow plant


@pytest.mark.gpu
def test_opt_controller_terminal_cost():
    """use_terminal_cost (opt_controller.py:84-98, 312-317): pyth_lq's own x'Px and a user-supplied torch function on
    pyth_idpendulum - cost and Jacobian against the oracle's rollout + autograd, shooting and collocation; the warning /
    assertion behaviour of the reference's constructor."""
    from gops_amd.create_pkg.create_env_model import create_env_model
    from gops_amd.sys_simulator.opt_controller import OptController
    from gops_amd.utils.synthetic import make_batch
    from oracle import adp_oracle as orc
    T, ci, gamma = 12, 3, 0.97
    user_tc = lambda s: 3.0 * (s[1:3] ** 2).sum() + 0.1 * torch.sin(s[0]) ** 2
    for cfg, tc in ((dict(env_id="pyth_lq", lq_config="s4a2"), None), (dict(env_id="pyth_idpendulum"), user_tc)):
        model = create_env_model(**cfg, use_gpu=True)
        env = orc.make_env(cfg["env_id"], lq_config=cfg.get("lq_config", "s4a2"))
        data = make_batch(dict(cfg, batch=2), 9)
        if tc is None:
            P = torch.as_tensor(model.unwrapped.dynamics.P, dtype=torch.float32)
            want_tc = lambda s: s @ P @ s
        else:
            want_tc = tc
        sho = OptController(model, num_pred_step=T, ctrl_interval=ci, gamma=gamma, mode="shooting", use_terminal_cost=True, terminal_cost=tc)
        plain = OptController(model, num_pred_step=T, ctrl_interval=ci, gamma=gamma, mode="shooting")
        rng = np.random.RandomState(2)
        for b in range(2):
            u = rng.uniform(0.5 * sho.bounds.lb, 0.5 * sho.bounds.ub)
            x = data["obs"][b].numpy()
            cost, jac = sho._cost_fcn_and_jac(u, x, {})
            acts = torch.tensor(u, dtype=torch.float32).reshape(T // ci, -1).repeat_interleave(ci, 0)[None]
            want_cost, _, want_jac = orc.raw_shooting_cost(env, data["obs"][b:b + 1], {}, acts, gamma, terminal=want_tc)
            assert abs(cost - want_cost.item()) <= 1e-4 * max(1.0, abs(want_cost.item())), (cfg, cost, want_cost)
            want = want_jac[0].reshape(T // ci, ci, -1).sum(1).reshape(-1)
            assert rel_l2(jac, want) < 1e-4, (cfg, rel_l2(jac, want))
            assert abs(cost - plain._cost_fcn_and_jac(u, x, {})[0]) > 1e-6      # the terminal term is really there
        # collocation: the terminal cost sits on the true final state of the last interval
        n, A, O = T // ci, sho.action_dim, sho.obs_dim
        col = OptController(model, num_pred_step=T, ctrl_interval=ci, gamma=gamma, use_terminal_cost=True, terminal_cost=tc)
        z = torch.tensor(rng.uniform(-0.3, 0.3, size=(n, A + O)), dtype=torch.float32)
        x0 = data["obs"][0]
        cost, jac = col._col_cost_and_jac(z.reshape(-1).numpy(), x0.numpy(), {})
        step = (lambda o, a: orc.lq_step(env["lq"], o, a)) if env["kind"] == "lq" else (lambda o, a: orc.idp_step(o, a))
        zf = z.reshape(-1).clone().requires_grad_(True)
        zz = zf.reshape(n, A + O)
        xs, want_cost = torch.cat((x0.reshape(1, O), zz[:-1, A:]), 0), torch.zeros(())
        for i in range(ci):
            xs, r, _ = step(xs, zz[:, :A])
            want_cost = want_cost - (r * gamma ** (torch.arange(n, dtype=torch.float32) * ci + i)).sum()
        want_cost = want_cost + want_tc(xs[-1]) * gamma ** T
        (want_jac,) = torch.autograd.grad(want_cost, zf)
        assert abs(cost - want_cost.item()) <= 1e-4 * max(1.0, abs(want_cost.item()))
        assert rel_l2(jac, want_jac) < 1e-4, (cfg, "collocation", rel_l2(jac, want_jac))
    with pytest.raises(AssertionError, match="no available terminal cost"):
        OptController(create_env_model("pyth_idpendulum", use_gpu=True), num_pred_step=4, use_terminal_cost=True)
    with pytest.warns(UserWarning, match="will be ignored"):
        OptController(create_env_model("pyth_idpendulum", use_gpu=True), num_pred_step=4, terminal_cost=user_tc)
    with pytest.raises(NotImplementedError):
        OptController(create_env_model("pyth_veh3dofconti", pre_horizon=10, use_gpu=True), num_pred_step=4, mode="shooting",
                      use_terminal_cost=True, terminal_cost=user_tc)


def _oracle_collocation(env, x, z, n, ci, A, O, gamma):
    """The reference's collocation rollout in batch mode (opt_controller.py:272-291, 196-215, 302-318) on the oracle's raw model
    steps: cost and transition residuals of the decision vector z [n, A + O], with autograd Jacobians."""
    import torch
    from oracle import adp_oracle as orc
    step = (lambda o, a: orc.lq_step(env["lq"], o, a)) if env["kind"] == "lq" else (lambda o, a: orc.idp_step(o, a))

    def both(zf):
        zz = zf.reshape(n, A + O)
        xs = torch.cat((x.reshape(1, O), zz[:-1, A:]), 0)
        us = zz[:, :A]
        cost = torch.zeros(())
        for i in range(ci):
            xs, r, _ = step(xs, us)
            k = torch.arange(n, dtype=torch.float32) * ci + i
            cost = cost - (r * gamma ** k).sum()
        return cost, (xs - zz[:, A:]).reshape(-1)

    zf = z.reshape(-1).clone().requires_grad_(True)
    cost, res = both(zf)
    (jac,) = torch.autograd.grad(cost, zf)
    J = torch.autograd.functional.jacobian(lambda v: both(v)[1], z.reshape(-1))
    return cost.item(), jac, res.detach(), J


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(env_id="pyth_lq", lq_config="s4a2"), dict(env_id="pyth_lq", lq_config="s6a3"), dict(env_id="pyth_idpendulum")],
                         ids=lambda c: c["env_id"][5:] + c.get("lq_config", ""))
def test_opt_controller_collocation_cost_constraints_and_jacobians(cfg):
    """mode="collocation" (the reference's default): cost + Jacobian w.r.t. (action, state) of every control point, transition
    residuals + their Jacobian - each from one or two kernel launches over the batch of intervals - against the reference's
    batch rollout restated on the oracle's raw model steps with autograd."""
    from gops_amd.create_pkg.create_env_model import create_env_model
    from gops_amd.sys_simulator.opt_controller import OptController
    from oracle import adp_oracle as orc
    model = create_env_model(**cfg, use_gpu=True)
    T, ci, gamma = 12, 3, 0.97
    ctrl = OptController(model, num_pred_step=T, ctrl_interval=ci, gamma=gamma)   # default mode
    assert ctrl.mode == "collocation" and ctrl.optimize_dim == ctrl.action_dim + ctrl.obs_dim
    env = orc.make_env(cfg["env_id"], lq_config=cfg.get("lq_config", "s4a2"))
    n, A, O = ctrl.num_ctrl_points, ctrl.action_dim, ctrl.obs_dim
    rng = np.random.RandomState(7)
    alo, ahi = ctrl.bounds.lb[:A], ctrl.bounds.ub[:A]
    for trial in range(3):
        x = rng.uniform(-0.3, 0.3, O).astype(np.float32)
        z = np.concatenate([np.concatenate((rng.uniform(0.5 * alo, 0.5 * ahi), rng.uniform(-0.3, 0.3, O))) for _ in range(n)]).astype(np.float32)
        want_cost, want_jac, want_res, want_J = _oracle_collocation(env, torch.tensor(x), torch.tensor(z), n, ci, A, O, gamma)
        cost, jac = ctrl._col_cost_and_jac(z, x, {})
        assert abs(cost - want_cost) <= 1e-4 * max(1.0, abs(want_cost)), (cost, want_cost)
        assert rel_l2(jac, want_jac) < 1e-4, rel_l2(jac, want_jac)
        res = ctrl._trans_constraint_fcn(z, x, {})
        assert rel_l2(res, want_res) < 1e-4, rel_l2(res, want_res)
        J = ctrl._trans_constraint_jac(z, x, {})
        assert J.shape == tuple(want_J.shape) and rel_l2(J, want_J) < 1e-4, rel_l2(J, want_J)


@pytest.mark.gpu
def test_opt_controller_collocation_and_shooting_agree_on_lq():
    """The two transcriptions of the same convex problem (pyth_lq s4a2, 16 steps, 8 control points) reach the same optimum: first
    action within 2e-3 of the action range, optimal costs within 1e-4; the collocation solution satisfies its dynamics."""
    from gops_amd.create_pkg.create_env_model import create_env_model
    from gops_amd.sys_simulator.opt_controller import OptController
    model = create_env_model("pyth_lq", lq_config="s4a2", use_gpu=True)
    x = np.array([1.0, -0.4, 0.7, 0.3], dtype=np.float32)
    col = OptController(model, num_pred_step=16, ctrl_interval=2, gamma=1.0, minimize_options={"maxiter": 300, "ftol": 1e-10})
    sho = OptController(model, num_pred_step=16, ctrl_interval=2, gamma=1.0, minimize_options={"maxiter": 300}, mode="shooting")
    u_col, u_sho = col(x), sho(x)
    res = col._trans_constraint_fcn(col.last_result.x, x, {})
    assert np.abs(res).max() < 1e-4, np.abs(res).max()
    rng_a = col.bounds.ub[:2] - col.bounds.lb[:2]
    assert np.all(np.abs(u_col - u_sho) < 2e-3 * rng_a), (u_col, u_sho)
    assert abs(col.last_result.fun - sho.last_result.fun) <= 1e-4 * max(1.0, abs(sho.last_result.fun)), (col.last_result.fun, sho.last_result.fun)
    # warm start keeps the (action, state) layout: shifted by one control point
    assert col.initial_guess.shape == (8 * 6,) and np.allclose(col.initial_guess[:6], col.last_result.x[6:12])


@pytest.mark.gpu
@pytest.mark.parametrize("case", [dict(B=100, sizes=[10, 64, 64, 20], act="gelu"), dict(B=4096, sizes=[46, 256, 256, 60], act="elu"),
                                  dict(B=17, sizes=[6, 32, 7], act="tanh"), dict(B=1, sizes=[4, 16, 48, 32, 130], act="relu"),
                                  # weight-gradient GEMM variants: LDS-ring kernel with an odd number of 16-sample tiles (B = 4000
                                  # -> 250 tiles, 272 -> 17), register-direct kernel with 128-tiles and ragged edges (192, 144 wide)
                                  # and with 64-tiles; the output layer's GEMM sees Wp = 144 / 16 / 256
                                  dict(B=4000, sizes=[30, 128, 256, 130], act="elu"), dict(B=272, sizes=[128, 256, 128, 3], act="relu"),
                                  dict(B=1000, sizes=[20, 192, 144, 256], act="selu"), dict(B=48, sizes=[16, 64, 96, 64, 5], act="sigmoid")],
                         ids=lambda c: f"B{c['B']}-{'x'.join(map(str, c['sizes']))}-{c['act']}")
def test_wide_output_mlp_matches_torch(case):
    """gops_mlp_forward / _backward (FiniteHorizonFullPolicy's evaluation in FHADP2: output width = act_dim * H, odd
    widths, ragged batches) against torch fp32 autograd: outputs and every parameter gradient at 1e-4 / 1e-5."""
    from gops_amd import hip_backend as hb
    from oracle import adp_oracle as orc
    torch.manual_seed(3)
    sizes, B = case["sizes"], case["B"]
    lin = [torch.nn.Linear(a, b) for a, b in zip(sizes[:-1], sizes[1:])]
    ws = [l.weight.detach().cuda().contiguous() for l in lin]
    bs = [l.bias.detach().cuda().contiguous() for l in lin]
    x = torch.randn(B, sizes[0])
    gy = torch.randn(B, sizes[-1])
    net = hb.MlpNet(hb.make_mlp(ws, bs, case["act"]), B)
    y = net.forward(x.cuda())
    gw, gb = [torch.empty_like(w) for w in ws], [torch.empty_like(b) for b in bs]
    net.backward(x.cuda(), gy.cuda().contiguous(), gw, gb)
    torch.cuda.synchronize()
    wr = [l.weight.detach().clone().requires_grad_(True) for l in lin]
    br = [l.bias.detach().clone().requires_grad_(True) for l in lin]
    yr = orc.mlp_forward(wr, br, x, case["act"])
    grads = torch.autograd.grad(yr, wr + br, grad_outputs=gy)
    assert rel_l2(y.cpu(), yr.detach()) < 1e-5
    for got, want in zip(gw + gb, grads):
        assert rel_l2(got.cpu(), want) < 1e-4, (case, tuple(want.shape), rel_l2(got.cpu(), want))


@pytest.mark.parametrize("name", ["spil_surrcstr_p10", "spil_detour_p8", "spil_errcstr_p10", "spil_veh2dof_errcstr_p10"])
def test_spil_class_matches_reference(name):
    """SPIL (create_alg surface) on the constrained veh3dofconti models: one full update - value and policy gradients,
    losses, safe probabilities and the PI multipliers - against the reference's, from its checkpoint layout."""
    alg, g, cfg = _load_alg(name)
    alg.gamma, alg.forward_step = cfg["gamma"], cfg["horizon"]
    alg.delta_i, alg.safe_prob_pre = np.array(g["state/delta_i"]), np.array(g["state/safe_prob_pre"])
    data = data_from_golden(g)
    tb, info = alg.get_remote_update_info(data, 0)
    assert list(info) == ["v", "policy"]
    assert abs(float(tb["Loss/Critic loss-RL iter"]) - float(g["pev_loss"])) <= TOL * max(1.0, abs(float(g["pev_loss"])))
    assert abs(float(tb["Train/Critic avg value-RL iter"]) - float(g["pev_vmean"])) <= TOL
    np.testing.assert_allclose(alg.safe_prob, g["safe_prob"], atol=1e-6)
    np.testing.assert_allclose(alg.lam, g["lam"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(alg.delta_i, g["after/delta_i"], rtol=1e-6, atol=1e-7)
    assert abs(float(tb["Loss/Actor loss-RL iter"]) - float(g["pim_loss"])) <= TOL * max(1.0, abs(float(g["pim_loss"])))
    for i, gr in enumerate(info["v"]):
        assert rel_l2(gr.cpu(), g[f"pev_grad/{i}"]) < TOL, ("pev", i)
    for i, gr in enumerate(info["policy"]):
        assert rel_l2(gr.cpu(), g[f"pim_grad/{i}"]) < TOL, ("pim", i, rel_l2(gr.cpu(), g[f"pim_grad/{i}"]))
    before = [p.detach().clone() for p in alg.networks.policy.parameters()]
    alg.remote_update(info)
    assert any((a - b).abs().max() > 0 for a, b in zip(alg.networks.policy.parameters(), before))


# ---- adjoint I/O around rollouts / MLP batches (ABI v5) and MPG ---------------------------------------------------------
@pytest.mark.parametrize("case", [dict(sizes=[5, 64, 64, 1], act="relu", B=70, params=True),
                                  dict(sizes=[4, 64, 64, 2], act="gelu", B=33, params=True),
                                  dict(sizes=[7, 32, 32, 1], act="tanh", B=16, params=False),
                                  dict(sizes=[37, 256, 256, 3], act="elu", B=300, params=False)])
def test_mlp_backward_x_matches_autograd(case):
    """`gops_mlp_backward_x`: d(loss)/d(x) of an MLP batch (with or without parameter gradients) against torch autograd
    of the oracle's restatement."""
    from oracle import adp_oracle as orc
    torch.manual_seed(5)
    sizes, B = case["sizes"], case["B"]
    lin = [torch.nn.Linear(a, b) for a, b in zip(sizes[:-1], sizes[1:])]
    ws = [l.weight.detach().cuda().contiguous() for l in lin]
    bs = [l.bias.detach().cuda().contiguous() for l in lin]
    x, gy = torch.randn(B, sizes[0]), torch.randn(B, sizes[-1])
    net = hb.MlpNet(hb.make_mlp(ws, bs, case["act"]), B)
    y = net.forward(x.cuda())
    gw, gb = ([torch.empty_like(w) for w in ws], [torch.empty_like(b) for b in bs]) if case["params"] else (None, None)
    gx = net.backward_x(x.cuda(), gy.cuda().contiguous(), gw, gb)
    torch.cuda.synchronize()
    xr = x.clone().requires_grad_(True)
    wr = [l.weight.detach().clone().requires_grad_(True) for l in lin]
    br = [l.bias.detach().clone().requires_grad_(True) for l in lin]
    yr = orc.mlp_forward(wr, br, xr, case["act"])
    grads = torch.autograd.grad(yr, [xr] + wr + br, grad_outputs=gy)
    assert rel_l2(y.cpu(), yr.detach()) < 1e-5
    assert rel_l2(gx.cpu(), grads[0]) < TOL, rel_l2(gx.cpu(), grads[0])
    if case["params"]:
        for got, want in zip(gw + gb, grads[1:]):
            assert rel_l2(got.cpu(), want) < TOL, (case, tuple(want.shape))


@pytest.mark.parametrize("env_kw,horizon,act,first_only", [
    (dict(env_id="pyth_lq", lq_config="s4a2"), 6, "gelu", True),
    (dict(env_id="pyth_lq", lq_config="s3a1"), 9, "elu", False),
    (dict(env_id="pyth_idpendulum"), 5, "tanh", True),
    (dict(env_id="gym_cartpoleconti"), 10, "relu", True),
    (dict(env_id="gym_pendulum"), 8, "elu", True),
    (dict(env_id="gym_pendulum"), 1, "relu", True),
])
def test_rollout_backward_adj_matches_autograd(env_kw, horizon, act, first_only):
    """`gops_rollout_backward_adj`: the sweep seeded with d(loss)/d(final_obs), returning d(loss)/d(obs_0), with the
    parameter gradients of step 0 only (later steps through a frozen copy of the policy, mpg.py:343-349) or of every
    step, against torch autograd over the oracle's rollout."""
    from helpers import hip_env_from_oracle, hip_mlp_from_net
    from oracle import adp_oracle as orc
    cfg = dict(env_kw, alg="INFADP", batch=37, horizon=horizon, hidden=(64, 64), act=act, gamma=0.97)
    env = orc.make_env(cfg["env_id"], lq_config=cfg.get("lq_config", "s4a2"))
    O, A, B = env["obs_dim"], env["act_dim"], cfg["batch"]
    net = orc.make_net([O, 64, 64, A], act, seed=11, act_high=torch.ones(A), act_low=-torch.ones(A))
    data = make_batch(cfg, 3)
    data["done"][-3:] = 1.0   # finished trajectories: MaskAtDone freezes the observation, the adjoint passes through
    gen = torch.Generator().manual_seed(9)
    gv, gfo = torch.randn(B, generator=gen) / B, torch.randn(B, O, generator=gen) / B
    # oracle: loss = sum_b gv_b v_b + <gfo, final_obs>
    obs0 = data["obs"].clone().requires_grad_(True)
    frozen = dict(net, w=[w.detach() for w in net["w"]], b=[b.detach() for b in net["b"]])
    o, done, info, v = obs0, data["done"].bool(), data, 0
    for t in range(horizon):
        a = orc.policy_forward(net if (t == 0 or not first_only) else frozen, o, None)
        o, r, done, info = orc.env_forward(env, o, a, done, info)
        v = v + r * cfg["gamma"] ** t
    loss = (gv * v).sum() + (gfo * o).sum()
    params = [p for pair in zip(net["w"], net["b"]) for p in pair]
    want = torch.autograd.grad(loss, [obs0] + params)
    # HIP
    dev = torch.device("cuda")
    mlp, ws, bs = hip_mlp_from_net(net, dev)
    ro = hb.Rollout(hip_env_from_oracle(env, net), mlp, batch=B, horizon=horizon, gamma=cfg["gamma"], finite_horizon=False)
    res = ro.forward({k: v_.to(dev).contiguous() for k, v_ in data.items() if k in ("obs", "done")}, want_final=True)
    gw, gb = [torch.empty_like(w) for w in ws], [torch.empty_like(b) for b in bs]
    g_obs = ro.backward_adj(gv.to(dev), gw, gb, grad_final_obs=gfo.to(dev).contiguous(), want_grad_obs=True,
                            first_step_only=first_only)
    torch.cuda.synchronize()
    assert rel_l2(res["final_obs"].cpu(), o.detach()) < 1e-5
    assert rel_l2(g_obs.cpu(), want[0]) < TOL, rel_l2(g_obs.cpu(), want[0])
    got = [p for pair in zip(gw, gb) for p in pair]
    for i, (g_, w_) in enumerate(zip(got, want[1:])):
        assert rel_l2(g_.cpu(), w_) < TOL, (i, rel_l2(g_.cpu(), w_))
    # no parameter gradients wanted: the input adjoint alone
    g_obs2 = ro.backward_adj(gv.to(dev), grad_final_obs=gfo.to(dev).contiguous(), want_grad_obs=True, first_step_only=first_only)
    assert torch.equal(g_obs2, g_obs)


@pytest.mark.parametrize("name", ["mpg_cartpole_mixed_weight", "mpg_pendulum_mixed_state", "mpg_lq_s4a2_mixed_weight",
                                  "mpg_idp_mixed_state"])
def test_mpg_class_matches_reference(name):
    """MPG (create_alg surface): one compute_gradient - the gradients of q1, q2 (q1_model, q2_model) and of the policy
    (data-driven + model-driven mix) and the logged scalars - against the reference's, from its checkpoint layout;
    then one update through the public API."""
    alg, g, cfg = _load_alg(name)
    meta = golden_meta(g)
    alg.gamma, alg.forward_step, alg.reward_scale = cfg["gamma"], cfg["horizon"], meta["reward_scale"]
    data = data_from_golden(g)
    tb, info = alg.get_remote_update_info(data, meta["iteration"])
    for k in [k for k in g if k.startswith("tb/")]:
        want = float(g[k])
        assert abs(float(tb[k[3:]]) - want) <= TOL * max(1.0, abs(want)), (k, tb[k[3:]], want)
    nets = [k[:-5] for k in info if k.endswith("_grad")]
    assert set(nets) == {k.split("/")[0][:-5] for k in g if "_grad/" in k}
    for n in nets:
        for i, gr in enumerate(info[f"{n}_grad"]):
            assert rel_l2(gr.cpu(), g[f"{n}_grad/{i}"]) < TOL, (n, i, rel_l2(gr.cpu(), g[f"{n}_grad/{i}"]))
    before = {n: [p.detach().clone() for p in getattr(alg.networks, n).parameters()] for n in nets + ["q1_target", "policy4rollout"]}
    alg.remote_update(info)
    for n in nets + ["q1_target", "policy4rollout"]:
        after = list(getattr(alg.networks, n).parameters())
        assert all(torch.isfinite(a).all() for a in after)
        assert any((a - b).abs().max() > 0 for a, b in zip(after, before[n])), n
    for a, b in zip(alg.networks.policy4rollout.parameters(), alg.networks.policy.parameters()):
        assert torch.equal(a, b)


def test_mpg_learns_with_device_sampler(tmp_path):
    """MPG end to end on the GPU: N pendulum models stepped by gops_env_step with exploration noise, device replay buffer,
    off_serial_trainer; the twin-Q regression loss falls and every network stays finite."""
    from gops_amd.create_pkg.create_buffer import create_buffer
    from gops_amd.create_pkg.create_trainer import create_trainer
    from gops_amd.trainer.sampler.device_env_sampler import DeviceEnvSampler
    cfg = dict(alg="MPG", env_id="gym_pendulum", batch=256, horizon=10, hidden=(64, 64), act="relu", gamma=0.99)
    torch.manual_seed(0)
    kw = _kwargs(cfg, dict(pge_method="mixed_weight", eta=0.3, terminal_iter=1e8, forward_step=10, tau=0.1), 0)
    kw.update(trainer="off_serial_trainer", buffer_name="replay_buffer", buffer_max_size=20000, buffer_warm_size=2048,
              replay_batch_size=256, sample_interval=1, additional_info={}, max_iteration=80, log_save_interval=1000,
              apprfunc_save_interval=1000, eval_interval=10 ** 9, save_folder=str(tmp_path), ini_network_dir=None)
    alg = create_alg(**kw)
    alg.reward_scale = 0.1
    alg.networks.to("cuda")
    smp = DeviceEnvSampler(cfg, alg.envmodel, n_envs=256, steps_per_sample=2, max_episode_steps=50, seed=3, noise_std=0.2,
                           env_step="model")
    # (the gym_cartpoleconti DATA env is restated too: its semantics are the sampler's default there)
    cp = DeviceEnvSampler(dict(cfg, env_id="gym_cartpoleconti"), create_alg(**dict(kw, env_id="gym_cartpoleconti", obsv_dim=4)).envmodel,
                          n_envs=32, steps_per_sample=3, max_episode_steps=50, seed=1)
    cp.networks = create_alg(**dict(kw, env_id="gym_cartpoleconti", obsv_dim=4)).networks.to("cuda")
    cb, _ = cp.sample()
    assert cb["obs"].abs().max() < 0.6 and bool((cb["rew"] == 1.0).all())   # reset at +-0.05, reward 1 per step
    buf = create_buffer(**kw)
    trainer = create_trainer(alg, smp, buf, None, **kw)
    losses = []
    for _ in range(80):
        trainer.step()
        trainer.iteration += 1
        losses.append(alg.tb_info["MPG/loss_q-RL iter"])
    assert all(np.isfinite(losses)) and np.mean(losses[-10:]) < 0.5 * np.mean(losses[:5]), (losses[:5], losses[-10:])
    assert all(torch.isfinite(p).all() for p in alg.networks.parameters())


@pytest.mark.parametrize("pge", ["mixed_weight", "mixed_state"])
def test_mpg_graph_replay_equals_eager(pge, monkeypatch):
    """local_update captured into a HIP graph (gradient + Adam per network + Polyak, iteration-dependent mixing weights
    as a graph input) walks exactly the parameter trajectory of the eager launches."""
    # (B = 256 for 30 iterations: the configuration in which memset NODES inside the captured graph - hipMemsetAsync in
    # the library - were seen to race with the neighbouring kernels; the library zero-fills with a kernel since)
    cfg = dict(alg="MPG", env_id="gym_pendulum", batch=256, horizon=10, hidden=(64, 64), act="relu", gamma=0.99)
    extra = dict(pge_method=pge, forward_step=10, tau=0.1, delay_update=2, **(dict(eta=0.3, terminal_iter=20) if pge == "mixed_weight" else dict(kappa=0.5)))
    B = cfg["batch"]

    def run(mode):
        monkeypatch.setenv("GOPS_HIP_GRAPH", mode)
        torch.manual_seed(4)
        alg = create_alg(**_kwargs(cfg, extra, 4))
        alg.networks.to("cuda")
        logs = []
        for it in range(30):
            gen = torch.Generator().manual_seed(100 + it)
            obs = make_batch(cfg, 60 + it)["obs"]
            data = dict(obs=obs, act=torch.rand(B, 1, generator=gen) * 2 - 1, rew=torch.randn(B, generator=gen),
                        obs2=obs + 0.05 * torch.randn(obs.shape, generator=gen), done=(torch.rand(B, generator=gen) < 0.1).float())
            logs.append(dict(alg.local_update(data, it)))
        return alg, logs

    eager, log_e = run("0")
    graph, log_g = run("1")
    assert any(c.graph is not None for c in graph._graphs.values()) and all(c.graph is None for c in eager._graphs.values())
    for (ne, pe), (ng, pg) in zip(eager.networks.named_parameters(), graph.networks.named_parameters()):
        assert ne == ng and torch.equal(pe, pg), ne
    for a, b in zip(log_e, log_g):
        for k in a:
            if not k.startswith("Time/"):
                assert a[k] == b[k], k


def test_mpg_gradient_is_the_mean_over_batch_shards():
    """Size-independent property at a BASELINE-sized batch (B = 4096, H = 30, 256-wide networks): with fixed mixing
    weights every MPG gradient is a batch mean, so the gradient of the whole batch equals the average of the gradients
    of its two halves (what the data-parallel trainer relies on)."""
    cfg = dict(alg="MPG", env_id="pyth_lq", lq_config="s4a2", batch=4096, horizon=30, hidden=(256, 256), act="elu", gamma=0.99)
    extra = dict(pge_method="mixed_weight", eta=0.2, terminal_iter=100, forward_step=30, tau=0.1)
    torch.manual_seed(2)
    alg = create_alg(**_kwargs(cfg, extra, 2))
    alg.networks.to("cuda")
    gen = torch.Generator().manual_seed(5)
    obs = make_batch(cfg, 9)["obs"]
    B, A = cfg["batch"], act_dim_of(cfg)
    data = dict(obs=obs, act=torch.rand(B, A, generator=gen) * 2 - 1, rew=torch.randn(B, generator=gen),
                obs2=obs + 0.05 * torch.randn(obs.shape, generator=gen), done=(torch.rand(B, generator=gen) < 0.1).float())

    def grads(sl):
        _, info = alg.get_remote_update_info({k: v[sl] for k, v in data.items()}, 40)
        return {k: [g.clone() for g in v] for k, v in info.items() if k.endswith("_grad")}

    whole, lo, hi = grads(slice(0, B)), grads(slice(0, B // 2)), grads(slice(B // 2, B))
    for k in whole:
        for i, g in enumerate(whole[k]):
            assert rel_l2(0.5 * (lo[k][i] + hi[k][i]).cpu(), g.cpu()) < 1e-4, (k, i, rel_l2(0.5 * (lo[k][i] + hi[k][i]).cpu(), g.cpu()))


@pytest.mark.gpu
def test_loss_scalars_and_polyak_update_match_torch():
    """ABI v11: gops_value_loss / gops_mean_loss / gops_polyak_update against the torch passes they replace
    (gops/algorithm/infadp.py:124-133, 172-173, 213), ragged sizes, repeated calls on one stats buffer, bit-reproducible."""
    from gops_amd import hip_backend as hb
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(3)
    st = hb.LossStats(dev)
    for n in (1, 77, 4096, 8192, 8193, 65536, 100003):   # (one block forms the sums up to 8192 elements)
        v = torch.randn(n, generator=g).to(dev) * 3 - 1
        b = torch.randn(n, generator=g).to(dev)
        grad = torch.empty(n, device=dev)
        out = st.value_loss(v, b, grad).clone()
        ref = torch.stack((((v - b).double() ** 2).mean(), v.double().mean())).float()
        assert torch.allclose(out, ref, rtol=2e-6, atol=1e-7), (n, out, ref)
        assert torch.equal(grad, (2.0 / n) * (v - b)) or torch.allclose(grad, (2.0 / n) * (v - b), rtol=2e-7, atol=0)
        again = st.value_loss(v, b, None).clone()
        assert torch.equal(out, again)
        m = st.mean_loss(v, -1.0).clone()
        assert torch.allclose(m, torch.stack((-v.double().mean(), v.double().mean())).float(), rtol=2e-6, atol=1e-7)
        assert float(st.buf[2:].abs().sum()) == 0.0 or st.buf[-2] == 0   # the ticket is left zero
    shapes = [(256, 126), (256,), (256, 256), (256,), (2, 256), (2,)] * 3   # 18 tensors: two chunks of the table
    online = [torch.randn(*s, generator=g).to(dev) for s in shapes]
    target = [torch.randn(*s, generator=g).to(dev) for s in shapes]
    want = [t.clone() for t in target]
    tau = 0.005
    torch._foreach_mul_(want, 1 - tau)
    torch._foreach_add_(want, online, alpha=tau)
    pk = hb.PolyakUpdater(target, online)
    assert pk.matches(target, online)
    pk.step(tau)
    for t, w in zip(target, want):
        assert torch.allclose(t, w, rtol=0, atol=1.2e-7 * float(w.abs().max()))   # (torch may fuse alpha * x + y into one fma)


def test_fused_update_tail_equals_separate_calls(monkeypatch):
    """ABI v12, `gops_rollout_backward_update`: loss mean + Adam step folded into the backward's last launch against the three separate
    calls (`gops_rollout_backward`, `gops_mean_loss`, `gops_adam_step`) - the same arithmetic element for element, so weights, Adam
    moments, step counts and the logged loss scalars are BIT-equal after several updates.  Shapes: a 64-wide net (exact fp32 kernels),
    the 256-wide plane-split kernels, and a batch beyond one block's loss-mean limit (8192: the mean then takes its own launch inside the
    same call)."""
    from gops_amd import hip_backend as hb
    from gops_amd.utils.synthetic import make_batch
    dev = torch.device("cuda", 0)
    cases = [dict(alg="FHADP", env_id="pyth_idpendulum", batch=96, horizon=8, hidden=(64, 64), act="gelu", gamma=0.99),
             dict(alg="FHADP", env_id="pyth_veh3dofconti", batch=200, horizon=6, pre_horizon=10, hidden=(256, 256), act="elu", gamma=1.0),
             dict(alg="FHADP", env_id="pyth_lq", lq_config="s4a2", batch=8192 + 40, horizon=3, hidden=(256, 256), act="relu", gamma=0.99)]
    for cfg in cases:
        def make():
            torch.manual_seed(7)
            alg = create_alg(**_kwargs(cfg, {}, 7), precision_check_interval=0)
            alg.networks.to(dev)
            return alg
        fused, plain = make(), make()
        monkeypatch.setattr(plain.networks.policy_optimizer, "begin_fused", lambda: None)   # -> backward, mean_loss, step() as three calls
        calls = []
        real = hb.lib().gops_rollout_backward_update
        for it in range(4):
            data = {k: v.to(dev) for k, v in make_batch(cfg, 40 + it).items()}
            tb_f = fused.local_update(data, it)
            tb_p = plain.local_update(data, it)
            assert float(tb_f[tb_tags["loss_actor"]]) == float(tb_p[tb_tags["loss_actor"]])
        torch.cuda.synchronize()
        assert fused.networks.policy_optimizer._fused is None and not hasattr(plain.networks.policy_optimizer, "_fused")
        for (name, a), b in zip(fused.networks.policy.named_parameters(), plain.networks.policy.parameters()):
            assert torch.equal(a, b), (cfg["env_id"], name)
            assert torch.equal(a.grad, b.grad), (cfg["env_id"], name)
            sa, sb = fused.networks.policy_optimizer.state[a], plain.networks.policy_optimizer.state[b]
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]) and int(sa["step"]) == int(sb["step"]) == 4
        # the device-resident step count / beta powers advanced alike
        df, dp = fused.networks.policy_optimizer._dev[0]["state"][0], plain.networks.policy_optimizer._dev[0]["state"][0]
        assert torch.equal(df, dp)
        moved = max((a - b).abs().max().item() for a, b in zip(make().networks.policy.parameters(), fused.networks.policy.parameters()))
        assert moved > 1e-4   # (the updates did move the weights)


def test_update_tail_is_refused_where_it_does_not_belong():
    """`GopsUpdateTail`: an Adam table that does not cover the policy's gradient tensors, or a tail on half a backward, is an error
    before anything is launched - never a silently skipped optimizer step."""
    from gops_amd import hip_backend as hb
    from gops_amd.utils.synthetic import make_batch
    dev = torch.device("cuda", 0)
    cfg = dict(alg="FHADP", env_id="pyth_lq", lq_config="s4a2", batch=64, horizon=3, hidden=(64, 64), act="elu", gamma=0.99)
    torch.manual_seed(3)
    alg = create_alg(**_kwargs(cfg, {}, 3), precision_check_interval=0)
    alg.networks.to(dev)
    data = {k: v.to(dev) for k, v in make_batch(cfg, 9).items()}
    alg.local_update(data, 0)   # (allocates gradients, optimizer state, workspace)
    opt = alg.networks.policy_optimizer
    before = [p.detach().clone() for p in alg.networks.policy.parameters()]
    ro = alg._rollout_for(64, dev)
    batch = alg._device_batch(data)
    v_pi = ro.forward(batch)["v_pi"]
    from gops_amd.algorithm.base import grad_buffers
    gw, gb = grad_buffers(alg.networks.policy)
    fa = opt.begin_fused()
    table = fa[0]
    table.n -= 1                                  # one tensor short
    with pytest.raises(RuntimeError, match="gops_rollout_backward_update"):
        ro.backward(alg._grad_v(64, dev), gw, gb, tail=hb.make_update_tail(fa, v_pi, -1.0, hb.LossStats(dev)))
    table.n += 1
    with pytest.raises(ValueError):
        ro.backward(alg._grad_v(64, dev), gw, gb, phase="a", tail=hb.make_update_tail(fa, v_pi, -1.0, hb.LossStats(dev)))
    opt._fused = None
    torch.cuda.synchronize()
    for a, b in zip(alg.networks.policy.parameters(), before):
        assert torch.equal(a, b)                  # nothing stepped


def test_fused_update_tail_infadp(monkeypatch):
    """The same for INFADP: the policy's backward carries loss mean + Adam + the Polyak step of policy_target (PIM,
    `gops_rollout_backward_update`), the value net's backward carries Adam + the Polyak step of v_target (PEV,
    `gops_value_backward_update`) - bit-equal networks, TARGETS and optimizer state after alternating PEV / PIM updates."""
    dev = torch.device("cuda", 0)
    cfg = dict(alg="INFADP", env_id="pyth_lq", lq_config="s4a2", batch=300, horizon=5, hidden=(256, 256), act="gelu", gamma=0.99)

    def make():
        torch.manual_seed(5)
        alg = create_alg(**_kwargs(cfg, {}, 5), precision_check_interval=0)
        alg.forward_step, alg.gamma = cfg["horizon"], cfg["gamma"]
        alg.networks.to(dev)
        return alg
    fused, plain = make(), make()
    monkeypatch.setenv("GOPS_HIP_GRAPH", "0")
    for net in ("policy", "v"):   # -> backward, mean_loss / step() / gops_polyak_update as separate calls
        monkeypatch.setattr(plain.networks.optimizer_dict[net], "begin_fused", lambda: None)
    for it in range(6):
        data = {k: v.to(dev) for k, v in make_batch(cfg, 70 + it).items()}
        tf, tp = fused.local_update(data, it), plain.local_update(data, it)
        for k in tf:
            if k != tb_tags["alg_time"]:
                assert float(tf[k]) == float(tp[k]), k
    torch.cuda.synchronize()
    names = [n for n, _ in fused.networks.named_parameters()]
    assert any("target" in n for n in names)   # (the comparison below covers the Polyak-averaged copies)
    for (name, a), b in zip(fused.networks.named_parameters(), plain.networks.parameters()):
        assert torch.equal(a, b), name
    init = make()
    assert max((a - b).abs().max().item() for (n, a), b in zip(init.networks.named_parameters(), fused.networks.parameters()) if "target" in n) > 1e-6
    for net in ("policy", "v"):
        of, op = fused.networks.optimizer_dict[net], plain.networks.optimizer_dict[net]
        for a, b in zip(fused.networks.net_dict[net].parameters(), plain.networks.net_dict[net].parameters()):
            assert torch.equal(of.state[a]["exp_avg"], op.state[b]["exp_avg"]) and int(of.state[a]["step"]) == int(op.state[b]["step"]) == 3
