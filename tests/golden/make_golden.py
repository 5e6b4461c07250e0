"""Generate the golden fixtures in this directory by running the UNMODIFIED reference.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
Every array written here is an output of the reference's own classes
(`gops.create_pkg.create_env_model.create_env_model`, `gops.algorithm.fhadp.FHADP`,
`gops.algorithm.infadp.INFADP`) on inputs from `gops_amd.utils.synthetic`.  The fixtures are what
pins `oracle/adp_oracle.py` and, through it (and directly), the HIP path.
"""
import json
import os
import sys
import zlib
from copy import deepcopy

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _ref_import  # noqa: E402

_ref_import.install()

from gops.algorithm.fhadp import FHADP  # noqa: E402
from gops.algorithm.fhadp2 import FHADP2  # noqa: E402
from gops.algorithm.fhadp_exterior import FHADPExterior  # noqa: E402
from gops.algorithm.fhadp_interior import FHADPInterior  # noqa: E402
from gops.algorithm.fhadp_lagrangian import FHADPLagrangian  # noqa: E402
from gops.algorithm.infadp import INFADP  # noqa: E402
from gops.create_pkg.create_env_model import create_env_model  # noqa: E402

from gops_amd.utils.synthetic import CONFIGS, act_dim_of, make_batch, obs_dim_of  # noqa: E402

torch.set_num_threads(8)


def alg_kwargs(cfg, seed, **extra):
    A = act_dim_of(cfg)
    kw = dict(
        algorithm=cfg["alg"], trainer="off_serial_trainer", seed=seed, cnn_shared=False,
        env_id=cfg["env_id"], obsv_dim=obs_dim_of(cfg), action_dim=A, action_type="continu",
        action_high_limit=np.ones(A, dtype=np.float32), action_low_limit=-np.ones(A, dtype=np.float32),
        policy_func_type="MLP",
        policy_func_name="FiniteHorizonPolicy" if cfg["alg"] == "FHADP" else "DetermPolicy",
        policy_hidden_sizes=list(cfg["hidden"]), policy_hidden_activation=cfg["act"],
        policy_act_distribution="default", policy_learning_rate=1e-3, use_gpu=False,
    )
    if cfg["alg"] in ("INFADP", "MAC"):
        kw.update(value_func_type="MLP", value_func_name="StateValue",
                  value_hidden_sizes=list(cfg["hidden"]), value_hidden_activation=cfg["act"],
                  value_learning_rate=1e-3)
    if "pre_horizon" in cfg or cfg["alg"] == "FHADP":
        kw["pre_horizon"] = cfg.get("pre_horizon", cfg["horizon"])
    if "lq_config" in cfg:
        kw["lq_config"] = cfg["lq_config"]
    if "surr_veh_num" in cfg:
        kw["surr_veh_num"] = cfg["surr_veh_num"]
    if cfg["alg"].startswith("FHADP") and cfg["alg"] not in ("FHADP", "FHADP2"):
        kw["policy_func_name"] = "FiniteHorizonPolicy"
        kw["pre_horizon"] = cfg.get("pre_horizon", cfg["horizon"])
    if cfg["alg"] == "FHADP2":
        kw["policy_func_name"] = "FiniteHorizonFullPolicy"
    kw.update(extra)
    return kw


def sd_to_np(sd, prefix="sd/"):
    return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items()}


def build_alg(cfg, seed, **extra):
    torch.manual_seed(seed)
    kw = alg_kwargs(cfg, seed, **extra)
    if cfg["alg"] == "FHADP":
        alg = FHADP(**kw)
        alg.gamma = cfg.get("gamma", 1.0)
    elif cfg["alg"] == "FHADP2":
        alg = FHADP2(**kw)
        alg.gamma = cfg.get("gamma", 1.0)
    elif cfg["alg"] in ("FHADPExterior", "FHADPInterior", "FHADPLagrangian"):
        cls = dict(FHADPExterior=FHADPExterior, FHADPInterior=FHADPInterior, FHADPLagrangian=FHADPLagrangian)[cfg["alg"]]
        alg = cls(**kw)
        alg.gamma = cfg.get("gamma", 1.0)
    elif cfg["alg"] == "MAC":
        from gops.algorithm.mac import MAC
        alg = MAC(**kw)
        alg.gamma = cfg.get("gamma", 0.99)
        alg.forward_step = cfg["horizon"]
    else:
        alg = INFADP(**kw)
        alg.gamma = cfg.get("gamma", 0.99)
        alg.forward_step = cfg["horizon"]
    return alg


def grads_of(module):
    return [p.grad.detach().clone() for p in module.parameters()]


def sample_idx(n, k, seed):
    return np.random.RandomState(seed).choice(n, size=min(k, n), replace=False)


def model_consts(model):
    """Constants the reference derives at construction with LAPACK (`torch.linalg.pinv` in fp32,
    lq_base.py:55-57): their last bits depend on the host CPU, so the fixtures carry the values the
    recorded outputs were computed with."""
    dyn = getattr(model.unwrapped, "dynamics", None)
    if dyn is not None and hasattr(dyn, "inv_IA"):
        return {"const/lq_inv_IA": dyn.inv_IA.detach().numpy().copy()}
    return {}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ------------------------------------------------------------------------------------------
# 1. single wrapped env-model steps (next_obs / reward / done; next state for veh3dof)
# ------------------------------------------------------------------------------------------
OBS_SCALE_STEP_CASES = {   # ScaleObservationModel (example_train/fhadp/fhadp_mlp_lqs3a1_serial.py: obs_scale [1, 2, 0.5]; lqs5a1: 10)
    "step_lq_s3a1_obsscale": (dict(env_id="pyth_lq", lq_config="s3a1"), dict(obs_scale=[1, 2, 0.5])),
    "step_lq_s5a1_obsscale_shift": (dict(env_id="pyth_lq", lq_config="s5a1"),
                                    dict(obs_scale=10, obs_shift=[0.1, -0.2, 0.05, 0.0, 0.3], reward_scale=0.5, reward_shift=1.0)),
    "step_idp_obsscale_shift": (dict(env_id="pyth_idpendulum"), dict(obs_scale=[1, 2, 2, 0.5, 0.5, 0.25], obs_shift=0.1)),
}
VEH2_STEP_CASES = {"step_veh2dof_p10": (dict(env_id="pyth_veh2dofconti", pre_horizon=10), {})}
VEH2_SMALL = {   # example_train/fhadp/fhadp_mlp_veh2dofconti_serial.py, infadp/infadp_mlp_veh2dofconti_offserial.py
    "fhadp_veh2dof_p10_elu": (dict(alg="FHADP", env_id="pyth_veh2dofconti", batch=48, horizon=10, pre_horizon=10,
                                   hidden=(64, 64), act="elu", gamma=1.0), {}),
    "infadp_veh2dof_p10_gelu": (dict(alg="INFADP", env_id="pyth_veh2dofconti", batch=40, horizon=8, pre_horizon=10,
                                     hidden=(64, 64), act="gelu", gamma=0.99), {}),
}
GYM_STEP_CASES = {   # gym-style models of the INFADP / MAC example scripts (cartpoleconti uses obs_scale in its script)
    "step_cartpole": (dict(env_id="gym_cartpoleconti"), {}),
    "step_cartpole_obsscale": (dict(env_id="gym_cartpoleconti"), dict(obs_scale=[1.0, 0.5, 2.0, 0.25], reward_scale=0.5, reward_shift=0.1)),
    "step_pendulum": (dict(env_id="gym_pendulum"), {}),
}
GYM_SMALL = {
    "infadp_cartpole_gelu": (dict(alg="INFADP", env_id="gym_cartpoleconti", batch=64, horizon=10,
                                  hidden=(64, 64), act="gelu", gamma=0.99), dict(obs_scale=[1.0, 1.0, 1.0, 1.0])),
    "mac_pendulum_elu": (dict(alg="MAC", env_id="gym_pendulum", batch=48, horizon=12,
                              hidden=(64, 64), act="elu", gamma=0.99), {}),
    "infadp_pendulum_tanh": (dict(alg="INFADP", env_id="gym_pendulum", batch=33, horizon=8,
                                  hidden=(64, 64), act="tanh", gamma=0.97), dict(reward_scale=0.1)),
}


ERR_STEP_CASES = {"step_veh_errcstr_p10": dict(env_id="pyth_veh3dofconti_errcstr", pre_horizon=10),
                  "step_veh2dof_errcstr_p10": dict(env_id="pyth_veh2dofconti_errcstr", pre_horizon=10)}
ERR_ALG_CASES = {
    "fhadp_ext_errcstr": (dict(alg="FHADPExterior", env_id="pyth_veh3dofconti_errcstr", batch=48, horizon=10, pre_horizon=10,
                               hidden=(64, 64), act="elu", gamma=0.99), dict(penalty=2.0)),
    "fhadp_int_veh2dof_errcstr": (dict(alg="FHADPInterior", env_id="pyth_veh2dofconti_errcstr", batch=40, horizon=10, pre_horizon=10,
                                       hidden=(64, 64), act="gelu", gamma=1.0), dict(penalty=1.5)),
    "fhadp_lag_errcstr": (dict(alg="FHADPLagrangian", env_id="pyth_veh3dofconti_errcstr", batch=40, horizon=8, pre_horizon=8,
                               hidden=(64, 64), act="tanh", gamma=1.0), dict(multiplier=0.6)),
}
SPIL_CASES = {   # gops/algorithm/spil.py on the constrained veh3dofconti models (one full update: PEV + PIM gradients)
    "spil_veh2dof_errcstr_p10": (dict(alg="SPIL", env_id="pyth_veh2dofconti_errcstr", batch=40, horizon=10, pre_horizon=10,
                                      hidden=(64, 64), act="elu", gamma=0.99), dict(constraint_dim=1)),
    "spil_errcstr_p10": (dict(alg="SPIL", env_id="pyth_veh3dofconti_errcstr", batch=48, horizon=10, pre_horizon=10,
                              hidden=(64, 64), act="relu", gamma=0.99), dict(constraint_dim=2)),
    "spil_surrcstr_p10": (dict(alg="SPIL", env_id="pyth_veh3dofconti_surrcstr", batch=48, horizon=10, pre_horizon=10,
                               hidden=(64, 64), act="elu", gamma=0.99), dict(constraint_dim=1)),
    "spil_detour_p8": (dict(alg="SPIL", env_id="pyth_veh3dofconti_detour", batch=40, horizon=8, pre_horizon=8,
                            hidden=(64, 64), act="gelu", gamma=0.97), dict(constraint_dim=3)),
}


def golden_spil():
    from gops.algorithm.spil import SPIL
    for name, (cfg, extra) in SPIL_CASES.items():
        seed = zlib.crc32(name.encode()) % 1000
        torch.manual_seed(seed)
        kw = alg_kwargs(dict(cfg, alg="INFADP"), seed, **extra)
        kw["algorithm"] = "SPIL"
        alg = SPIL(gamma=cfg["gamma"], forward_step=cfg["horizon"], **kw)
        perturb_targets(alg, seed)
        data = make_batch(cfg, seed)
        data["done"][-4:] = 1.0
        data["constraint"] = torch.zeros(cfg["batch"], extra["constraint_dim"])
        # a non-trivial multiplier state, as after some updates
        alg.delta_i = np.array([3.0, 1.0, 0.5][:extra["constraint_dim"]])
        alg.safe_prob_pre = np.array([0.9, 0.8, 0.95][:extra["constraint_dim"]])
        out = {"in/" + k: v.numpy().copy() for k, v in data.items()}
        out["meta/cfg"] = json.dumps(dict(cfg=cfg, extra=extra, seed=seed))
        out["state/delta_i"], out["state/safe_prob_pre"] = alg.delta_i.copy(), alg.safe_prob_pre.copy()
        out.update(sd_to_np(alg.networks.state_dict()))
        tb, info = alg.get_remote_update_info(data, 0)
        for i, gr in enumerate(info["v"]):
            out[f"pev_grad/{i}"] = gr.detach().numpy().copy()
        for i, gr in enumerate(info["policy"]):
            out[f"pim_grad/{i}"] = gr.detach().numpy().copy()
        out["pev_loss"], out["pev_vmean"] = tb["Loss/Critic loss-RL iter"], tb["Train/Critic avg value-RL iter"]
        out["pim_loss"] = tb["Loss/Actor loss-RL iter"]
        out["safe_prob"], out["lam"] = np.asarray(alg.safe_prob), np.asarray(alg.lam)
        out["after/delta_i"] = alg.delta_i.copy()
        save(name, **out)


# MultiRefTrajModel(path_para, u_para): every path and both speed profiles away from their defaults
REFPARA = dict(path_para={"sine": {"A": 2.0, "omega": 0.5, "phi": 0.3},
                          "double_lane": {"t1": 4.0, "t2": 8.5, "t3": 13.0, "t4": 17.5, "y1": 0.5, "y2": 3.0},
                          "triangle": {"A": 2.5, "T": 8.0}, "circle": {"r": 80.0}},
               u_para={"sine": {"A": 1.5, "omega": 0.7, "phi": 0.2, "b": 6.0}, "constant": {"u": 4.0}})
REFPARA_STEP_CASES = {"step_veh_p10_refpara": (dict(env_id="pyth_veh3dofconti", pre_horizon=10), REFPARA),
                      "step_veh2dof_p10_refpara": (dict(env_id="pyth_veh2dofconti", pre_horizon=10), REFPARA)}
REFPARA_SMALL = {"fhadp_veh_p10_refpara": (dict(alg="FHADP", env_id="pyth_veh3dofconti", batch=48, horizon=10, pre_horizon=10,
                                                hidden=(64, 64), act="elu", gamma=1.0), REFPARA)}

# ActionRepeatModel (repeat_num / sum_reward of create_env_model; example_train/fhadp/fhadp_mlp_idpendulum_serial.py:41)
REPEAT_STEP_CASES = {
    "step_idp_repeat3": (dict(env_id="pyth_idpendulum"), dict(repeat_num=3)),
    "step_lq_s3a1_repeat2_last_obsscale": (dict(env_id="pyth_lq", lq_config="s3a1"),
                                           dict(repeat_num=2, sum_reward=False, obs_scale=[1, 2, 0.5], reward_scale=0.5, reward_shift=1.0)),
    "step_cartpole_repeat4": (dict(env_id="gym_cartpoleconti"), dict(repeat_num=4)),
    "step_pendulum_repeat2": (dict(env_id="gym_pendulum"), dict(repeat_num=2, sum_reward=False)),
}
REPEAT_SMALL = {
    "fhadp_idp_repeat2_gelu": (dict(alg="FHADP", env_id="pyth_idpendulum", batch=40, horizon=8, hidden=(64, 64), act="gelu",
                                    gamma=1.0), dict(repeat_num=2)),
    "infadp_lq_s4a2_repeat3_elu": (dict(alg="INFADP", env_id="pyth_lq", lq_config="s4a2", batch=48, horizon=6, hidden=(64, 64),
                                        act="elu", gamma=0.99), dict(repeat_num=3, sum_reward=False)),
    "fhadp_pendulum_repeat3_tanh": (dict(alg="FHADP", env_id="gym_pendulum", batch=33, horizon=7, hidden=(64, 64), act="tanh",
                                         gamma=0.98), dict(repeat_num=3)),
    "infadp_cartpole_repeat2_relu": (dict(alg="INFADP", env_id="gym_cartpoleconti", batch=40, horizon=8, hidden=(64, 64),
                                          act="relu", gamma=0.99), dict(repeat_num=2)),
}

# create_env_model(mask_at_done=False): no MaskAtDoneModel in the chain
NOMASK_STEP_CASES = {"step_idp_nomask": (dict(env_id="pyth_idpendulum"), dict(mask_at_done=False)),
                     "step_veh_p10_nomask": (dict(env_id="pyth_veh3dofconti", pre_horizon=10), dict(mask_at_done=False)),
                     "step_cartpole_nomask_repeat2": (dict(env_id="gym_cartpoleconti"), dict(mask_at_done=False, repeat_num=2))}
NOMASK_SMALL = {
    "fhadp_veh_p10_nomask_elu": (dict(alg="FHADP", env_id="pyth_veh3dofconti", batch=48, horizon=10, pre_horizon=10,
                                      hidden=(64, 64), act="elu", gamma=1.0), dict(mask_at_done=False)),
    "infadp_cartpole_nomask_relu": (dict(alg="INFADP", env_id="gym_cartpoleconti", batch=64, horizon=12, hidden=(64, 64),
                                         act="relu", gamma=0.99), dict(mask_at_done=False)),
    "infadp_veh2dof_nomask_gelu": (dict(alg="INFADP", env_id="pyth_veh2dofconti", batch=40, horizon=8, pre_horizon=10,
                                        hidden=(64, 64), act="gelu", gamma=0.99), dict(mask_at_done=False)),
}

MPG_CASES = {   # gops/algorithm/mpg.py: one compute_gradient (twin-Q regression + mixed policy gradient) per case
    "mpg_cartpole_mixed_weight": (dict(alg="MPG", env_id="gym_cartpoleconti", batch=64, horizon=10, hidden=(64, 64), act="relu",
                                       gamma=0.99), dict(pge_method="mixed_weight", eta=0.3, terminal_iter=10000), 3000, 0.1),
    "mpg_pendulum_mixed_state": (dict(alg="MPG", env_id="gym_pendulum", batch=48, horizon=8, hidden=(64, 64), act="elu",
                                      gamma=0.99), dict(pge_method="mixed_state", kappa=0.5), 7, 0.1),
    "mpg_lq_s4a2_mixed_weight": (dict(alg="MPG", env_id="pyth_lq", lq_config="s4a2", batch=40, horizon=6, hidden=(64, 64),
                                      act="gelu", gamma=0.97), dict(pge_method="mixed_weight", eta=0.1, terminal_iter=100), 80, 1.0),
    "mpg_idp_mixed_state": (dict(alg="MPG", env_id="pyth_idpendulum", batch=33, horizon=5, hidden=(32, 32), act="tanh",
                                 gamma=0.99), dict(pge_method="mixed_state", kappa=0.15), 0, 0.5),
}


def golden_mpg():
    from gops.algorithm.mpg import MPG
    for name, (cfg, extra, iteration, reward_scale) in MPG_CASES.items():
        seed = zlib.crc32(name.encode()) % 1000
        torch.manual_seed(seed)
        kw = alg_kwargs(dict(cfg, alg="INFADP"), seed, **extra)
        kw.update(algorithm="MPG", value_func_name="ActionValue", value_output_activation="linear")
        alg = MPG(gamma=cfg["gamma"], forward_step=cfg["horizon"], tau=0.1, **kw)
        alg.reward_scale = reward_scale
        g = torch.Generator().manual_seed(seed + 1000)
        nets = alg.networks
        moved = [nets.q1_target, nets.q2_target, nets.policy_target]
        if extra["pge_method"] == "mixed_state":   # the model pair starts as a copy of (q1, q2): move it apart
            moved += [nets.q1_model, nets.q2_model, nets.q1_model_target, nets.q2_model_target]
        with torch.no_grad():
            for net in moved:
                for p in net.parameters():
                    p.add_(0.1 * (torch.rand(p.shape, generator=g) - 0.5))
        B, A = cfg["batch"], act_dim_of(cfg)
        obs = make_batch(cfg, seed)["obs"]
        data = dict(obs=obs, act=torch.rand(B, A, generator=g) * 2 - 1, rew=torch.randn(B, generator=g),
                    obs2=obs + 0.05 * torch.randn(obs.shape, generator=g), done=(torch.rand(B, generator=g) < 0.15).float())
        out = {"in/" + k: v.numpy().copy() for k, v in data.items()}
        out["meta/cfg"] = json.dumps(dict(cfg=cfg, extra=extra, seed=seed, iteration=iteration, reward_scale=reward_scale))
        out.update(sd_to_np(nets.state_dict()))
        out.update(model_consts(alg.envmodel))
        tb, info = alg.get_remote_update_info(data, iteration)
        for key, grads in info.items():
            if key.endswith("_grad"):
                for i, gr in enumerate(grads):
                    out[f"{key}/{i}"] = gr.detach().numpy().copy()
        for k, v in tb.items():
            if k.startswith("MPG/"):
                out["tb/" + k] = np.float64(v)
        save(name, **out)


MAC_SMALL = {   # gops/algorithm/mac.py on the info-free models
    "mac_lq_s4a2_gelu": (dict(alg="MAC", env_id="pyth_lq", lq_config="s4a2", batch=64, horizon=10,
                              hidden=(64, 64), act="gelu", gamma=0.99), {}),
    "mac_idp_elu": (dict(alg="MAC", env_id="pyth_idpendulum", batch=40, horizon=8,
                         hidden=(64, 64), act="elu", gamma=0.97), {}),
}
OBS_SCALE_SMALL = {
    "fhadp_lq_s3a1_obsscale": (dict(alg="FHADP", env_id="pyth_lq", lq_config="s3a1", batch=40, horizon=20,
                                    hidden=(64, 64), act="elu", gamma=0.99), dict(obs_scale=[1, 2, 0.5])),
    "infadp_lq_s5a1_obsscale_shift": (dict(alg="INFADP", env_id="pyth_lq", lq_config="s5a1", batch=48, horizon=8,
                                           hidden=(64, 64), act="gelu", gamma=0.99),
                                      dict(obs_scale=10, obs_shift=[0.1, -0.2, 0.05, 0.0, 0.3])),
    "fhadp_idp_obsscale_shift": (dict(alg="FHADP", env_id="pyth_idpendulum", batch=33, horizon=12,
                                      hidden=(64, 64), act="tanh", gamma=1.0),
                                 dict(obs_scale=[1, 2, 2, 0.5, 0.5, 0.25], obs_shift=0.1, reward_scale=0.1)),
}


def golden_steps(cases=None):
    cases = cases or {
        "step_lq_s4a2": (dict(env_id="pyth_lq", lq_config="s4a2"), {}),
        "step_lq_s6a3": (dict(env_id="pyth_lq", lq_config="s6a3"), {}),
        "step_lq_s2a1_shaped": (dict(env_id="pyth_lq", lq_config="s2a1"),
                                dict(reward_scale=0.5, reward_shift=1.0)),
        "step_idp": (dict(env_id="pyth_idpendulum"), dict(reward_scale=1)),
        "step_veh_p10": (dict(env_id="pyth_veh3dofconti", pre_horizon=10), {}),
        "step_veh_p30": (dict(env_id="pyth_veh3dofconti", pre_horizon=30), {}),
    }
    for name, (cfg, extra) in cases.items():
        B, nsteps = 48, 6
        data = make_batch(dict(cfg, batch=B), seed=7)
        model = create_env_model(**cfg, **extra)
        g = torch.Generator().manual_seed(11)
        A = act_dim_of(cfg)
        # a third of the batch starts done; observations far enough out to hit obs clipping (lq)
        done = (torch.rand(B, generator=g) < 0.3).float()
        obs = data["obs"].clone()
        if cfg["env_id"] == "pyth_lq":
            obs[:8] *= 8.0
        if cfg["env_id"] == "pyth_idpendulum":
            obs[:6, 1] = 1.2  # tips over -> done from the model
        info = {k: v.clone() for k, v in data.items()}
        out = {"in/obs": obs.numpy().copy(), "in/done": done.numpy().copy()}
        for k in ("state", "ref_points", "path_num", "u_num", "ref_time"):
            if k in data:
                out["in/" + k] = data[k].numpy().copy()
        o, d = obs, done
        for s in range(nsteps):
            a = torch.rand(B, A, generator=g) * 2.6 - 1.3  # beyond [-1,1]: exercises the clamps
            o, r, d, info = model.forward(o, a, d, info)
            out[f"s{s}/act"] = a.numpy()
            out[f"s{s}/obs"] = o.numpy().copy()
            out[f"s{s}/rew"] = r.numpy().copy()
            out[f"s{s}/done"] = d.numpy().copy()
            if "ref_points" in info and info.get("state") is not None:
                out[f"s{s}/state"] = info["state"].numpy().copy()
                out[f"s{s}/ref_last"] = info["ref_points"][:, -1].numpy().copy()
        out["meta/nsteps"] = nsteps
        out.update(model_consts(model))
        out["meta/cfg"] = json.dumps(dict(cfg=cfg, extra=extra))
        save(name, **out)


# ------------------------------------------------------------------------------------------
# 2. small FHADP / INFADP gradient cases with full inputs, weights and gradients
# ------------------------------------------------------------------------------------------
SMALL = {
    "fhadp_lq_s4a2_tanh": (dict(alg="FHADP", env_id="pyth_lq", lq_config="s4a2", batch=40, horizon=12,
                                hidden=(64, 64), act="tanh", gamma=0.97), {}),
    "fhadp_lq_s6a3_relu": (dict(alg="FHADP", env_id="pyth_lq", lq_config="s6a3", batch=33, horizon=8,
                                hidden=(32, 64, 32), act="relu", gamma=1.0), {}),
    "fhadp_idp_gelu": (dict(alg="FHADP", env_id="pyth_idpendulum", batch=64, horizon=10,
                            hidden=(64, 64), act="gelu", gamma=1.0), dict(reward_scale=1)),
    "fhadp_idp_selu_shaped": (dict(alg="FHADP", env_id="pyth_idpendulum", batch=24, horizon=16,
                                   hidden=(64, 64), act="selu", gamma=0.95),
                              dict(reward_scale=0.1, reward_shift=-2.0)),
    "fhadp_veh_p10_elu": (dict(alg="FHADP", env_id="pyth_veh3dofconti", batch=48, horizon=10,
                               pre_horizon=10, hidden=(64, 64), act="elu", gamma=1.0), {}),
    "fhadp_veh_p30_sigmoid": (dict(alg="FHADP", env_id="pyth_veh3dofconti", batch=20, horizon=30,
                                   pre_horizon=30, hidden=(128, 128), act="sigmoid", gamma=0.99), {}),
    "infadp_lq_s4a2_gelu": (dict(alg="INFADP", env_id="pyth_lq", lq_config="s4a2", batch=64, horizon=10,
                                 hidden=(64, 64), act="gelu", gamma=0.99), {}),
    "infadp_idp_gelu": (dict(alg="INFADP", env_id="pyth_idpendulum", batch=40, horizon=10,
                             hidden=(64, 64), act="gelu", gamma=0.99), {}),
    "infadp_veh_p10_relu": (dict(alg="INFADP", env_id="pyth_veh3dofconti", batch=32, horizon=10,
                                 pre_horizon=10, hidden=(64, 64, 64), act="relu", gamma=0.99), {}),
}


def perturb_targets(alg, seed):
    """Targets equal the online nets right after construction; move them so that tests can tell
    V from V_target (as after a few Polyak updates)."""
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for p in alg.networks.v_target.parameters():
            p.add_(0.05 * (torch.rand(p.shape, generator=g) - 0.5))


# the collision-penalty model trains with plain FHADP (example_train/fhadp/fhadp_mlp_veh3dofconti_surrcstr_penalty_serial.py)
PENALTY_SMALL = {
    "fhadp_surrpen_p10_elu": (dict(alg="FHADP", env_id="pyth_veh3dofconti_surrcstr_penalty", batch=48, horizon=10,
                                   pre_horizon=10, hidden=(64, 64), act="elu", gamma=1.0), {}),
    "fhadp_surrpen_p25_gelu": (dict(alg="FHADP", env_id="pyth_veh3dofconti_surrcstr_penalty", batch=33, horizon=25,
                                    pre_horizon=25, hidden=(128, 128), act="gelu", gamma=0.98), {}),
}


def golden_small(cases=None):
    for name, (cfg, extra) in (SMALL if cases is None else cases).items():
        seed = zlib.crc32(name.encode()) % 1000
        alg = build_alg(cfg, seed, **extra)
        data = make_batch(cfg, seed)
        if "idp" in name:
            data["obs"][:5, 1] = 0.9  # some trajectories fall over within the horizon
            data["obs2"] = data["obs"].clone()
        if cfg["env_id"] == "pyth_veh3dofconti":
            data["state"][:3, 1] += 9.3  # |delta_y| crosses 10 m within the horizon -> done
            from gops_amd.utils.synthetic import veh_obs_f32
            data["obs"] = torch.from_numpy(veh_obs_f32(data["state"].numpy(), data["ref_points"].numpy()))
            data["obs2"] = data["obs"].clone()
        if cfg["env_id"] == "pyth_veh2dofconti":
            data["state"][:4, 0] += 1.6  # |delta_y| crosses 2 m within the horizon -> done
            st, rp = data["state"], data["ref_points"]
            data["obs"] = torch.cat((st[:, :2] - rp[:, 0], st[:, 2:], st[:, :1] - rp[:, 1:, 0]), dim=1)
            data["obs2"] = data["obs"].clone()
        data["done"][-3:] = 1.0  # already-done rows in the batch
        out = {"in/" + k: v.numpy().copy() for k, v in data.items()}
        out["meta/cfg"] = json.dumps(dict(cfg=cfg, extra=extra, seed=seed))
        out.update(model_consts(alg.envmodel))
        if cfg["alg"] == "FHADP":
            out.update(sd_to_np(alg.networks.state_dict()))
            alg._compute_gradient(data)
            for i, gr in enumerate(grads_of(alg.networks.policy)):
                out[f"grad/{i}"] = gr.numpy()
            out["loss"] = alg.tb_info["Loss/Actor loss-RL iter"]
        else:
            perturb_targets(alg, seed)
            out.update(sd_to_np(alg.networks.state_dict()))
            _, info = alg.get_remote_update_info(data, 0)  # PEV
            for i, gr in enumerate(info["v"]):
                out[f"pev_grad/{i}"] = gr.detach().numpy().copy()
            out["pev_loss"] = alg.tb_info["Loss/Critic loss-RL iter"]
            out["pev_vmean"] = alg.tb_info["Train/Critic avg value-RL iter"]
            _, info = alg.get_remote_update_info(data, 1)  # PIM
            for i, gr in enumerate(info["policy"]):
                out[f"pim_grad/{i}"] = gr.detach().numpy().copy()
            out["pim_loss"] = alg.tb_info["Loss/Actor loss-RL iter"]
        save(name, **out)


# ------------------------------------------------------------------------------------------
# 3. BASELINE.json shapes: inputs/weights are regenerated from the seed by the tests, the
#    fixture holds checksums of them plus loss, per-parameter gradient norms and samples.
# ------------------------------------------------------------------------------------------
def golden_big():
    for name, cfg in CONFIGS.items():
        seed = 0
        alg = build_alg(cfg, seed)
        data = make_batch(cfg, seed)
        out = {"chk/obs_sum": data["obs"].double().sum().item()}
        out.update(model_consts(alg.envmodel))
        params0 = [p.detach().clone() for p in alg.networks.policy.parameters()]
        out["chk/policy_w0_sum"] = params0[0].double().sum().item()
        out["chk/policy_wlast_sum"] = params0[-2].double().sum().item()

        def pack(prefix, grads):
            out[prefix + "norms"] = np.array([g.double().norm().item() for g in grads])
            for i, g in enumerate(grads):
                flat = g.reshape(-1)
                idx = sample_idx(flat.numel(), 256, 100 + i)
                out[f"{prefix}idx{i}"] = idx
                out[f"{prefix}val{i}"] = flat[idx].numpy()

        if cfg["alg"] == "FHADP":
            alg._compute_gradient(data)
            pack("grad/", grads_of(alg.networks.policy))
            out["loss"] = alg.tb_info["Loss/Actor loss-RL iter"]
        else:
            perturb_targets(alg, seed)
            out["chk/vt_w0_sum"] = next(alg.networks.v_target.parameters()).double().sum().item()
            _, info = alg.get_remote_update_info(data, 0)
            pack("pev_grad/", [g.detach() for g in info["v"]])
            out["pev_loss"] = alg.tb_info["Loss/Critic loss-RL iter"]
            _, info = alg.get_remote_update_info(data, 1)
            pack("pim_grad/", [g.detach() for g in info["policy"]])
            out["pim_loss"] = alg.tb_info["Loss/Actor loss-RL iter"]
        save("big_" + name, **out)


# ------------------------------------------------------------------------------------------
# 4. the reference's shipped TRAINED checkpoints (results/<ALG>/<run>/apprfunc/*.pkl): saturating
#    policies, long horizons, non-unit action limits - same arrays as the small cases.
# ------------------------------------------------------------------------------------------
REF_ROOT = "/root/reference"
TRAINED = {
    "fhadp_trained_idp_h80": ("FHADP/idpendulum", "apprfunc_54000_opt.pkl", dict(batch=96)),
    "fhadp_trained_lqs3a1_h80": ("FHADP/lqs3a1", "apprfunc_5400_opt.pkl", dict(batch=80)),
    "infadp_trained_lqs4a2": ("INFADP/lqs4a2_mlp", "apprfunc_6000.pkl", dict(batch=128, horizon=10)),
    "infadp_trained_idp": ("INFADP/idpendulum", "apprfunc_90000_opt.pkl", dict(batch=96, horizon=10)),
}


def golden_trained():
    for name, (run, ckpt, over) in TRAINED.items():
        rc = json.load(open(os.path.join(REF_ROOT, "results", run, "config.json")))
        cfg = dict(alg=rc["algorithm"], env_id=rc["env_id"], hidden=tuple(rc["policy_hidden_sizes"]),
                   act=rc["policy_hidden_activation"], batch=over["batch"],
                   horizon=over.get("horizon", rc.get("pre_horizon")),
                   gamma=1.0 if rc["algorithm"] == "FHADP" else 0.99)
        if "lq_config" in rc:
            cfg["lq_config"] = rc["lq_config"]
        if rc["algorithm"] == "FHADP":
            cfg["pre_horizon"] = rc["pre_horizon"]
        extra = {k: rc[k] for k in ("reward_scale", "reward_shift") if rc.get(k) is not None}
        lim = dict(action_high_limit=np.array(rc["action_high_limit"], dtype=np.float32),
                   action_low_limit=np.array(rc["action_low_limit"], dtype=np.float32))
        seed = zlib.crc32(name.encode()) % 1000
        alg = build_alg(cfg, seed, **extra, **lim)
        sd = torch.load(os.path.join(REF_ROOT, "results", run, "apprfunc", ckpt), map_location="cpu")
        alg.networks.load_state_dict(sd)
        data = make_batch(cfg, seed)
        data["done"][-2:] = 1.0
        out = {"in/" + k: v.numpy().copy() for k, v in data.items()}
        out["meta/cfg"] = json.dumps(dict(cfg=cfg, extra=extra, seed=seed, checkpoint=f"results/{run}/apprfunc/{ckpt}"))
        out.update(sd_to_np(alg.networks.state_dict()))
        out.update(model_consts(alg.envmodel))
        if cfg["alg"] == "FHADP":
            alg._compute_gradient(data)
            for i, gr in enumerate(grads_of(alg.networks.policy)):
                out[f"grad/{i}"] = gr.numpy()
            out["loss"] = alg.tb_info["Loss/Actor loss-RL iter"]
        else:
            _, info = alg.get_remote_update_info(data, 0)  # PEV
            for i, gr in enumerate(info["v"]):
                out[f"pev_grad/{i}"] = gr.detach().numpy().copy()
            out["pev_loss"] = alg.tb_info["Loss/Critic loss-RL iter"]
            out["pev_vmean"] = alg.tb_info["Train/Critic avg value-RL iter"]
            _, info = alg.get_remote_update_info(data, 1)  # PIM
            for i, gr in enumerate(info["policy"]):
                out[f"pim_grad/{i}"] = gr.detach().numpy().copy()
            out["pim_loss"] = alg.tb_info["Loss/Actor loss-RL iter"]
        save(name, **out)


# ------------------------------------------------------------------------------------------
# 4b. 256-wide networks TRAINED by the unmodified reference (`FHADP._local_update`, fhadp.py:87-90;
#     `INFADP.local_update`, infadp.py:101-104, incl. its Polyak step): a few hundred Adam updates on
#     fresh synthetic batches, then one gradient evaluation on a held-out batch.  The BASELINE shapes
#     are all 256-wide, i.e. they run the plane-split (16-bit operand planes) kernels; the shipped
#     checkpoints above are 64-wide.  These fixtures pin the 1e-4 bar on weights that have MOVED
#     (larger magnitudes, a saturating tanh head in the *_sat case).
# ------------------------------------------------------------------------------------------
TRAINED256 = {
    # name: (cfg of the evaluation batch, training: updates / batch / learning rate)
    "t256_fhadp_idp_h30_gelu": (dict(alg="FHADP", env_id="pyth_idpendulum", batch=208, horizon=30, hidden=(256, 256),
                                     act="gelu", gamma=1.0), dict(updates=300, batch=256, lr=1e-3)),
    "t256_fhadp_veh_p30_elu": (dict(alg="FHADP", env_id="pyth_veh3dofconti", batch=176, horizon=30, pre_horizon=30,
                                    hidden=(256, 256), act="elu", gamma=1.0), dict(updates=300, batch=128, lr=1e-3)),
    # a learning rate high enough that the tanh head saturates (80 - 95 % of the actions beyond 0.99) - on pyth_lq, where the
    # closed loop stays well-conditioned (the same experiment on veh3dofconti ends in a chaotic bang-bang policy whose fp32
    # gradient scatters by 7e-2 under 1-ulp weight moves: not a parity fixture)
    "t256_fhadp_lq_s4a2_elu_sat": (dict(alg="FHADP", env_id="pyth_lq", lq_config="s4a2", batch=192, horizon=30,
                                        hidden=(256, 256), act="elu", gamma=0.99), dict(updates=300, batch=256, lr=3e-3)),
    "t256_infadp_lq_s4a2_relu": (dict(alg="INFADP", env_id="pyth_lq", lq_config="s4a2", batch=240, horizon=10,
                                      hidden=(256, 256), act="relu", gamma=0.99), dict(updates=400, batch=256, lr=1e-3)),
    "t256_infadp_lq_s4a2_gelu": (dict(alg="INFADP", env_id="pyth_lq", lq_config="s4a2", batch=240, horizon=10,
                                      hidden=(256, 256), act="gelu", gamma=0.99), dict(updates=400, batch=256, lr=1e-3)),
    # the shape of BASELINE configs[2] (three 256-wide relu layers + tail value net: the thinnest parity margin)
    "t256_infadp_veh_p10_relu3": (dict(alg="INFADP", env_id="pyth_veh3dofconti", batch=144, horizon=10, pre_horizon=10,
                                       hidden=(256, 256, 256), act="relu", gamma=0.99), dict(updates=300, batch=128, lr=1e-3)),
}


def ref_fp32_scatter(alg, net_names, grad_fn, trials=4):
    """How far the REFERENCE's own fp32 gradient moves when every weight of `net_names` moves by at most one ulp: the largest
    rel-L2 distance of `grad_fn()` (flat gradient) to its unperturbed value over `trials` seeded perturbations.  Near a trained
    optimum the mean gradient is a small difference of large per-trajectory terms, and this scatter reaches 1e-4 (veh3dofconti):
    a fixture's parity bar cannot be tighter than the noise of the thing it is compared with."""
    base = grad_fn().double()
    params = [p for n in net_names for p in getattr(alg.networks, n).parameters()]
    saved = [p.detach().clone() for p in params]
    gen = torch.Generator().manual_seed(0)
    worst = 0.0
    for _ in range(trials):
        with torch.no_grad():
            for p, s0 in zip(params, saved):
                p.copy_(s0 * (1 + (torch.rand(s0.shape, generator=gen) - 0.5) * 1.2e-7))
        worst = max(worst, float((grad_fn().double() - base).norm() / base.norm()))
    with torch.no_grad():
        for p, s0 in zip(params, saved):
            p.copy_(s0)
    return worst


def golden_trained256(only=None):
    for name, (cfg, tr) in TRAINED256.items():
        if only and name not in only:
            continue
        seed = zlib.crc32(name.encode()) % 1000
        alg = build_alg(cfg, seed, policy_learning_rate=tr["lr"], value_learning_rate=tr["lr"])
        w0 = torch.cat([p.detach().reshape(-1) for p in alg.networks.policy.parameters()]).clone()
        curve = []
        for it in range(tr["updates"]):
            batch = make_batch(cfg, 10_000 + 7 * seed + it, batch=tr["batch"])
            if cfg["alg"] == "FHADP":
                alg._local_update(batch, it)
                curve.append(alg.tb_info["Loss/Actor loss-RL iter"])
            else:
                alg.local_update(batch, it)      # PEV and PIM by `iteration % (pev_step + pim_step)`, Polyak inside
                curve.append(alg.tb_info.get("Loss/Actor loss-RL iter", np.nan))
        w1 = torch.cat([p.detach().reshape(-1) for p in alg.networks.policy.parameters()])
        data = make_batch(cfg, seed)
        data["done"][-2:] = 1.0
        out = {"in/" + k: v.numpy().copy() for k, v in data.items()}
        out["meta/cfg"] = json.dumps(dict(cfg=cfg, extra={}, seed=seed, training=tr))
        out["meta/loss_curve"] = np.asarray(curve, dtype=np.float64)
        out["meta/policy_moved_rel"] = float((w1 - w0).norm() / w0.norm())
        out["meta/policy_absmax"] = float(w1.abs().max())
        out.update(sd_to_np(alg.networks.state_dict()))
        out.update(model_consts(alg.envmodel))
        if cfg["alg"] == "FHADP":
            alg._compute_gradient(data)
            for i, gr in enumerate(grads_of(alg.networks.policy)):
                out[f"grad/{i}"] = gr.numpy()
            out["loss"] = alg.tb_info["Loss/Actor loss-RL iter"]
            with torch.no_grad():   # how saturated the tanh head is on the evaluation batch (|a| of the first step)
                a0 = alg.networks.policy(data["obs"], 1)
            out["meta/act0_absmean"] = float(a0.abs().mean())
            out["meta/act0_sat_share"] = float((a0.abs() > 0.99).float().mean())

            def flat_grad():
                alg._compute_gradient(data)
                return torch.cat([gr.reshape(-1) for gr in grads_of(alg.networks.policy)])
            out["meta/ref_fp32_scatter"] = ref_fp32_scatter(alg, ["policy"], flat_grad)
        else:
            _, info = alg.get_remote_update_info(data, 0)  # PEV
            for i, gr in enumerate(info["v"]):
                out[f"pev_grad/{i}"] = gr.detach().numpy().copy()
            out["pev_loss"] = alg.tb_info["Loss/Critic loss-RL iter"]
            out["pev_vmean"] = alg.tb_info["Train/Critic avg value-RL iter"]
            _, info = alg.get_remote_update_info(data, 1)  # PIM
            for i, gr in enumerate(info["policy"]):
                out[f"pim_grad/{i}"] = gr.detach().numpy().copy()
            out["pim_loss"] = alg.tb_info["Loss/Actor loss-RL iter"]

            def flat_pev():
                return torch.cat([gr.detach().reshape(-1) for gr in alg.get_remote_update_info(data, 0)[1]["v"]])

            def flat_pim():
                return torch.cat([gr.detach().reshape(-1) for gr in alg.get_remote_update_info(data, 1)[1]["policy"]])
            out["meta/ref_fp32_scatter_pev"] = ref_fp32_scatter(alg, ["v"], flat_pev)
            out["meta/ref_fp32_scatter_pim"] = ref_fp32_scatter(alg, ["policy", "v_target"], flat_pim)
        print(f"{name}: policy moved {out['meta/policy_moved_rel']:.3f} (rel L2), max|w| {out['meta/policy_absmax']:.3f}, "
              f"loss {curve[0]:.4g} -> {curve[-1]:.4g}" + (f", |a0| mean {out['meta/act0_absmean']:.3f}, saturated share "
              f"{out['meta/act0_sat_share']:.3f}, reference fp32 scatter {out['meta/ref_fp32_scatter']:.2e}" if cfg["alg"] == "FHADP" else
              f", reference fp32 scatter PEV {out['meta/ref_fp32_scatter_pev']:.2e} PIM {out['meta/ref_fp32_scatter_pim']:.2e}"))
        save(name, **out)


# ------------------------------------------------------------------------------------------
# 5. FHADP2 (open-loop action sequence from one FiniteHorizonFullPolicy evaluation)
# ------------------------------------------------------------------------------------------
FHADP2_CASES = {
    "fhadp2_lq_s4a2_tanh": (dict(alg="FHADP2", env_id="pyth_lq", lq_config="s4a2", batch=40, horizon=12,
                                 pre_horizon=12, hidden=(64, 64), act="tanh", gamma=0.97), {}),
    "fhadp2_idp_gelu": (dict(alg="FHADP2", env_id="pyth_idpendulum", batch=48, horizon=20, pre_horizon=20,
                             hidden=(64, 64), act="gelu", gamma=1.0), dict(reward_scale=1)),
    "fhadp2_veh_p10_elu": (dict(alg="FHADP2", env_id="pyth_veh3dofconti", batch=48, horizon=10, pre_horizon=10,
                                hidden=(128, 128), act="elu", gamma=1.0), {}),
}


def golden_fhadp2():
    for name, (cfg, extra) in FHADP2_CASES.items():
        seed = zlib.crc32(name.encode()) % 1000
        alg = build_alg(cfg, seed, **extra)
        data = make_batch(cfg, seed)
        if "idp" in name:
            data["obs"][:5, 1] = 0.9
            data["obs2"] = data["obs"].clone()
        if cfg["env_id"] == "pyth_veh3dofconti":
            data["state"][:3, 1] += 9.3
            from gops_amd.utils.synthetic import veh_obs_f32
            data["obs"] = torch.from_numpy(veh_obs_f32(data["state"].numpy(), data["ref_points"].numpy()))
            data["obs2"] = data["obs"].clone()
        data["done"][-3:] = 1.0
        out = {"in/" + k: v.numpy().copy() for k, v in data.items()}
        out["meta/cfg"] = json.dumps(dict(cfg=cfg, extra=extra, seed=seed))
        out.update(model_consts(alg.envmodel))
        out.update(sd_to_np(alg.networks.state_dict()))
        _, info = alg.get_remote_update_info(data, 0)
        for i, gr in enumerate(info["grad"]):
            out[f"grad/{i}"] = gr.detach().numpy().copy()
        out["loss"] = alg.tb_info["Loss/Actor loss-RL iter"]
        save(name, **out)


# ------------------------------------------------------------------------------------------
# 7. constrained FHADP variants on the veh3dofconti models with surrounding vehicles: model steps (obs / reward / done /
#    surr_state / constraint) and loss terms + gradients of FHADPExterior / FHADPInterior / FHADPLagrangian
# ------------------------------------------------------------------------------------------
CSTR_STEP_CASES = {
    "step_veh_surrcstr_p10": dict(env_id="pyth_veh3dofconti_surrcstr", pre_horizon=10),
    "step_veh_detour_p10": dict(env_id="pyth_veh3dofconti_detour", pre_horizon=10),
    "step_veh_surrcstr_p5_n2": dict(env_id="pyth_veh3dofconti_surrcstr", pre_horizon=5, surr_veh_num=2),
}
CSTR_ALG_CASES = {
    "fhadp_ext_surrcstr": (dict(alg="FHADPExterior", env_id="pyth_veh3dofconti_surrcstr", batch=48, horizon=10, pre_horizon=10,
                                hidden=(64, 64), act="elu", gamma=1.0), dict(penalty=2.5)),
    "fhadp_int_surrcstr": (dict(alg="FHADPInterior", env_id="pyth_veh3dofconti_surrcstr", batch=48, horizon=10, pre_horizon=10,
                                hidden=(64, 64), act="gelu", gamma=0.98), dict(penalty=1.7)),
    "fhadp_lag_surrcstr": (dict(alg="FHADPLagrangian", env_id="pyth_veh3dofconti_surrcstr", batch=40, horizon=8, pre_horizon=8,
                                hidden=(64, 64), act="tanh", gamma=1.0), dict(multiplier=0.8)),
    "fhadp_int_detour": (dict(alg="FHADPInterior", env_id="pyth_veh3dofconti_detour", batch=48, horizon=12, pre_horizon=12,
                              hidden=(64, 64), act="elu", gamma=1.0), dict(penalty=1.3)),
    "fhadp_ext_detour": (dict(alg="FHADPExterior", env_id="pyth_veh3dofconti_detour", batch=33, horizon=9, pre_horizon=9,
                              hidden=(64, 128), act="relu", gamma=0.99), dict(penalty=4.0)),
}


PENALTY_STEP_CASES = {"step_veh_surrpen_p10": dict(env_id="pyth_veh3dofconti_surrcstr_penalty", pre_horizon=10)}
# the penalty model also fills info["constraint"] (with the constraint of the CURRENT pose): the constrained classes run on it
PENALTY_ALG_CASES = {
    "fhadp_ext_surrpen": (dict(alg="FHADPExterior", env_id="pyth_veh3dofconti_surrcstr_penalty", batch=40, horizon=10, pre_horizon=10,
                               hidden=(64, 64), act="elu", gamma=0.99), dict(penalty=3.0)),
    "fhadp_int_surrpen": (dict(alg="FHADPInterior", env_id="pyth_veh3dofconti_surrcstr_penalty", batch=40, horizon=8, pre_horizon=8,
                               hidden=(64, 64), act="tanh", gamma=1.0), dict(penalty=1.5)),
}


def golden_constrained(step_cases=None, alg_cases=None):
    for name, cfg in (CSTR_STEP_CASES if step_cases is None else step_cases).items():
        B, nsteps = 48, 6
        data = make_batch(dict(cfg, batch=B), seed=19)
        model = create_env_model(**cfg)
        g = torch.Generator().manual_seed(23)
        done = (torch.rand(B, generator=g) < 0.25).float()
        info = {k: v.clone() for k, v in data.items()}
        out = {"in/" + k: data[k].numpy().copy() for k in ("obs", "state", "ref_points", "path_num", "u_num", "ref_time", "surr_state")
               if k in data}
        out["in/done"] = done.numpy().copy()
        o, d = data["obs"].clone(), done
        for s in range(nsteps):
            a = torch.rand(B, act_dim_of(cfg), generator=g) * 2.6 - 1.3
            o, r, d, info = model.forward(o, a, d, info)
            out[f"s{s}/act"] = a.numpy()
            out[f"s{s}/obs"], out[f"s{s}/rew"], out[f"s{s}/done"] = o.numpy().copy(), r.numpy().copy(), d.numpy().copy()
            out[f"s{s}/state"] = info["state"].numpy().copy()
            out[f"s{s}/ref_last"] = info["ref_points"][:, -1].numpy().copy()
            if "surr_state" in info:
                out[f"s{s}/surr_state"] = info["surr_state"].numpy().copy()
            out[f"s{s}/constraint"] = info["constraint"].numpy().copy()
        out["meta/nsteps"] = nsteps
        out["meta/cfg"] = json.dumps(dict(cfg=cfg, extra={}))
        save(name, **out)
    for name, (cfg, extra) in (CSTR_ALG_CASES if alg_cases is None else alg_cases).items():
        seed = zlib.crc32(name.encode()) % 1000
        alg = build_alg(cfg, seed, **extra)
        data = make_batch(cfg, seed)
        data["done"][-4:] = 1.0
        out = {"in/" + k: v.numpy().copy() for k, v in data.items()}
        out["meta/cfg"] = json.dumps(dict(cfg=cfg, extra=extra, seed=seed))
        out.update(sd_to_np(alg.networks.state_dict()))
        alg.networks.policy.zero_grad()
        loss, info = alg._compute_loss_policy(deepcopy(data))
        loss.backward()
        for i, gr in enumerate(grads_of(alg.networks.policy)):
            out[f"grad/{i}"] = gr
        out["loss"] = loss.item()
        for k, v in info.items():
            out["tb/" + k] = float(v)
        save(name, **out)


# ------------------------------------------------------------------------------------------
# 5b. pyth_mobilerobot (example_train/spil/spil_mlp_mobilerobot_*.py): the model draws np.random.normal inside every
#     forward (Robot.f_xu); the draws of the obstacle robot are recorded next to the outputs they produced
# ------------------------------------------------------------------------------------------
class record_normal:
    """Records every np.random.normal call made inside the block: list of (scale, float32 draws)."""

    def __enter__(self):
        self.calls, self._orig = [], np.random.normal

        def normal(loc=0.0, scale=1.0, size=None):
            out = self._orig(loc, scale, size)
            self.calls.append((float(scale), np.asarray(out, dtype=np.float32).copy()))
            return out
        np.random.normal = normal
        return self

    def __exit__(self, *exc):
        np.random.normal = self._orig

    def obstacle_draws(self):
        """[n_forward_calls, B, 2]: per model.forward the (v, w) draws of the obstacle (the ego's std-0 draws are zeros)."""
        obs = [d for sc, d in self.calls if sc != 0.0]
        assert len(obs) % 2 == 0 and len(self.calls) == 2 * len(obs)
        return np.stack([np.stack((obs[i], obs[i + 1]), axis=1) for i in range(0, len(obs), 2)])


MOB_ALG_CASES = {
    "spil_mobilerobot": (dict(alg="SPIL", env_id="pyth_mobilerobot", batch=48, horizon=10, hidden=(64, 64), act="relu", gamma=0.99),
                         dict(constraint_dim=1)),
    "fhadp_ext_mobilerobot": (dict(alg="FHADPExterior", env_id="pyth_mobilerobot", batch=40, horizon=8, hidden=(64, 64), act="elu",
                                   gamma=0.98), dict(penalty=3.0)),
    "infadp_mobilerobot_gelu": (dict(alg="INFADP", env_id="pyth_mobilerobot", batch=40, horizon=8, hidden=(64, 64), act="gelu",
                                     gamma=0.99), {}),
}


def golden_mobilerobot():
    cfg = dict(env_id="pyth_mobilerobot")
    B, nsteps = 48, 6
    np.random.seed(3)
    data = make_batch(dict(cfg, batch=B), seed=29)
    model = create_env_model(**cfg)
    g = torch.Generator().manual_seed(31)
    done = (torch.rand(B, generator=g) < 0.25).float()
    obs = data["obs"].clone()
    obs[:4, 0] = 59.96           # x passes the observation bound (60) within a step or two: ClipObservation acts
    obs[4:8, 1] = 3.9            # |y| > 4: done from the model
    out = {"in/obs": obs.numpy().copy(), "in/done": done.numpy().copy()}
    o, d, info = obs, done, {}
    for s in range(nsteps):
        a = torch.rand(B, 2, generator=g) * 2.6 - 1.3
        with record_normal() as rec:
            o, r, d, info = model.forward(o, a, d, info)
        out[f"s{s}/act"], out[f"s{s}/noise"] = a.numpy(), rec.obstacle_draws()[0]
        out[f"s{s}/obs"], out[f"s{s}/rew"], out[f"s{s}/done"] = o.numpy().copy(), r.numpy().copy(), d.numpy().copy()
        out[f"s{s}/constraint"] = info["constraint"].numpy().copy()
    out["meta/nsteps"] = nsteps
    out["meta/cfg"] = json.dumps(dict(cfg=cfg, extra={}))
    save("step_mobilerobot", **out)

    from gops.algorithm.spil import SPIL
    for name, (cfg, extra) in MOB_ALG_CASES.items():
        seed = zlib.crc32(name.encode()) % 1000
        np.random.seed(seed)
        data = make_batch(cfg, seed)
        data["done"][-4:] = 1.0
        if cfg["alg"] == "SPIL":
            torch.manual_seed(seed)
            kw = alg_kwargs(dict(cfg, alg="INFADP"), seed, **extra)
            kw["algorithm"] = "SPIL"
            alg = SPIL(gamma=cfg["gamma"], forward_step=cfg["horizon"], **kw)
            perturb_targets(alg, seed)
            data["constraint"] = torch.zeros(cfg["batch"], 1)
            alg.delta_i, alg.safe_prob_pre = np.array([3.0]), np.array([0.9])
            out = {"in/" + k: v.numpy().copy() for k, v in data.items()}
            out["state/delta_i"], out["state/safe_prob_pre"] = alg.delta_i.copy(), alg.safe_prob_pre.copy()
            out.update(sd_to_np(alg.networks.state_dict()))
            with record_normal() as rec:
                tb, info = alg.get_remote_update_info(data, 0)
            draws = rec.obstacle_draws()
            H = cfg["horizon"]
            assert draws.shape == (2 * H, cfg["batch"], 2)
            out["in/noise_pev"], out["in/noise_pim"] = draws[:H], draws[H:]
            for i, gr in enumerate(info["v"]):
                out[f"pev_grad/{i}"] = gr.detach().numpy().copy()
            for i, gr in enumerate(info["policy"]):
                out[f"pim_grad/{i}"] = gr.detach().numpy().copy()
            out["pev_loss"], out["pev_vmean"] = tb["Loss/Critic loss-RL iter"], tb["Train/Critic avg value-RL iter"]
            out["pim_loss"] = tb["Loss/Actor loss-RL iter"]
            out["safe_prob"], out["lam"] = np.asarray(alg.safe_prob), np.asarray(alg.lam)
            out["after/delta_i"] = alg.delta_i.copy()
        elif cfg["alg"] == "FHADPExterior":
            alg = build_alg(cfg, seed, **extra)
            out = {"in/" + k: v.numpy().copy() for k, v in data.items()}
            out.update(sd_to_np(alg.networks.state_dict()))
            alg.networks.policy.zero_grad()
            with record_normal() as rec:
                loss, info = alg._compute_loss_policy(deepcopy(data))
            out["in/noise"] = rec.obstacle_draws()
            loss.backward()
            for i, gr in enumerate(grads_of(alg.networks.policy)):
                out[f"grad/{i}"] = gr
            out["loss"] = loss.item()
            for k, v in info.items():
                out["tb/" + k] = float(v)
        else:   # INFADP: policy evaluation (iteration 0) and policy improvement (iteration 1), each with its own draws
            alg = build_alg(cfg, seed, **extra)
            perturb_targets(alg, seed)
            out = {"in/" + k: v.numpy().copy() for k, v in data.items()}
            out.update(sd_to_np(alg.networks.state_dict()))
            with record_normal() as rec:
                _, info = alg.get_remote_update_info(data, 0)
            out["in/noise_pev"] = rec.obstacle_draws()
            for i, gr in enumerate(info["v"]):
                out[f"pev_grad/{i}"] = gr.detach().numpy().copy()
            out["pev_loss"] = alg.tb_info["Loss/Critic loss-RL iter"]
            out["pev_vmean"] = alg.tb_info["Train/Critic avg value-RL iter"]
            with record_normal() as rec:
                _, info = alg.get_remote_update_info(data, 1)
            out["in/noise_pim"] = rec.obstacle_draws()
            for i, gr in enumerate(info["policy"]):
                out[f"pim_grad/{i}"] = gr.detach().numpy().copy()
            out["pim_loss"] = alg.tb_info["Loss/Actor loss-RL iter"]
        out["meta/cfg"] = json.dumps(dict(cfg=cfg, extra=extra, seed=seed))
        save(name, **out)


# ------------------------------------------------------------------------------------------
# 6. DATA-environment transitions (the numpy envs the reference's samplers step: create_env + its wrappers)
#    and the reference ReplayBuffer run on them - the semantics DeviceEnvSampler / the device ReplayBuffer
#    reproduce (terminal penalty -100, data-env termination tests, no observation clipping)
# ------------------------------------------------------------------------------------------
DATA_ENV_CASES = {
    "dataenv_veh_p10": dict(env_id="pyth_veh3dofconti", pre_horizon=10),
    "dataenv_lq_s4a2": dict(env_id="pyth_lq", lq_config="s4a2"),
    "dataenv_idp": dict(env_id="pyth_idpendulum"),
    "dataenv_lq_s2a1_shaped": dict(env_id="pyth_lq", lq_config="s2a1", reward_scale=0.5, reward_shift=1.0),
    "dataenv_cartpole": dict(env_id="gym_cartpoleconti", reward_scale=0.5),
    "dataenv_veh2dof_p10": dict(env_id="pyth_veh2dofconti", pre_horizon=10),
    # pyth_mobilerobot.py:108-152: Robot.f_xu with the heading CLIPPED to +-pi (the model does not clip), obstacle driven by its own
    # (v, w) + np.random.normal draws (recorded per transition: t/noise), no terminal penalty, the model's done test
    "dataenv_mobilerobot": dict(env_id="pyth_mobilerobot"),
}


def golden_data_envs(only=None):
    from gops.create_pkg.create_env import create_env
    from gops.trainer.buffer.replay_buffer import ReplayBuffer as RefBuffer
    for _old, _new in (("float_", np.float64), ("int_", np.int64), ("bool8", np.bool_)):   # reference targets numpy 1.x
        if not hasattr(np, _old):
            setattr(np, _old, _new)
    for name, cfg in DATA_ENV_CASES.items():
        if only is not None and name not in only:
            continue
        env = create_env(**cfg)
        rng = np.random.RandomState(zlib.crc32(name.encode()) % 10000)
        rows = []
        info_keys = ("state", "ref_points", "path_num", "u_num", "ref_time")
        episodes, n_done = 0, 0
        while len(rows) < 400:
            episodes += 1
            env.seed(int(rng.randint(1 << 30)))
            reset_kw = {}
            if cfg["env_id"] == "pyth_mobilerobot" and episodes % 4 != 1:
                # the reset distribution alone (robot x in [0, 2.7], obstacle x in [3.5, 6]) never reaches a termination or the heading clip
                # inside 40 steps: every second episode starts on a collision course, on the |y| = 4 edge, or with a heading next to pi
                reset_kw = dict(init_state=[
                    [3.0, 0.1, 0.0, 0.3, 0.0, 0, 0, 0, 3.9, 0.2, 3.1, 0.4, 0.0],
                    [1.0, 3.9, 1.5, 0.3, 0.0, 0, 0, 0, 5.0, -2.0, 1.6, 0.2, 0.0],
                    [1.0, 0.0, 3.05, 0.2, 0.9, 0, 0, 0, 5.0, 2.0, -3.1, 0.3, -0.9]][episodes % 4 - 2 if episodes % 4 >= 2 else 2])
            ret = env.reset(**reset_kw)
            obs, info = ret if isinstance(ret, tuple) else (ret, getattr(env, "info", {}))
            if not info:
                info = getattr(env.unwrapped, "info", {}) or {}
            # half of the episodes use violent actions so that terminations (and the -100) are in the fixture
            amp = 1.3 if episodes % 2 else 0.4
            for t in range(40):
                act = rng.uniform(-amp, amp, size=env.action_space.shape).astype(np.float32)
                cur_info = {k: np.array(info[k], dtype=np.float32).copy() for k in info_keys if k in info}
                extra_cols = {}
                if cfg["env_id"] == "pyth_mobilerobot":
                    with record_normal() as rec:
                        step = env.step(act)
                    extra_cols = dict(noise=rec.obstacle_draws()[0][0], constraint=np.float32(np.asarray(step[-1]["constraint"]).reshape(-1)[0]))
                else:
                    step = env.step(act)
                obs2, rew, done, info2 = step[0], step[1], step[2], step[-1]
                rows.append(dict(obs=np.array(obs, np.float32), act=act, rew=np.float32(rew), done=np.float32(np.asarray(done).reshape(-1)[0]),
                                 obs2=np.array(obs2, np.float32), **extra_cols,
                                 **{"info_" + k: v for k, v in cur_info.items()},
                                 **{"next_" + k: np.array(info2[k], dtype=np.float32) for k in cur_info}))
                obs, info = obs2, info2
                if done:
                    n_done += 1
                    break
        out = {"t/" + k: np.stack([r[k] for r in rows]) for k in rows[0]}
        print(name, "transitions", len(rows), "episodes", episodes, "terminations", n_done)
        out["meta/cfg"] = json.dumps(dict(cfg=cfg, extra={k: cfg[k] for k in ("reward_scale", "reward_shift") if k in cfg}))
        model = create_env_model(**cfg)
        out.update(model_consts(model))
        # the reference ReplayBuffer fed with the first 150 transitions (capacity 100: wraps), then two sampled batches
        add_info = {}
        for k in rows[0]:
            if k.startswith("info_"):
                add_info[k[5:]] = {"shape": rows[0][k].shape, "dtype": np.float32}
        bkw = dict(trainer="off_serial_trainer", seed=5, obsv_dim=rows[0]["obs"].shape[0], action_dim=rows[0]["act"].shape[0],
                   buffer_max_size=100, additional_info=add_info)
        buf = RefBuffer(index=0, **bkw)
        for r in rows[:150]:
            info = {k[5:]: r[k] for k in r if k.startswith("info_")}
            nxt = {k[5:]: r[k] for k in r if k.startswith("next_")}
            buf.store(r["obs"], r["act"], float(r["rew"]), bool(r["done"]), info, r["obs2"], nxt, 0.25)
        out["buf/size"], out["buf/ptr"] = buf.size, buf.ptr
        for k, v in buf.buf.items():
            out["buf/store/" + k] = np.asarray(v)
        out["meta/buffer_kwargs"] = json.dumps({k: (v if k != "additional_info" else {kk: {"shape": list(vv["shape"])} for kk, vv in v.items()})
                                                for k, v in bkw.items()})
        save(name, **out)


# ------------------------------------------------------------------------------------------
# 9. the reference's PrioritizedReplayBuffer (gops/trainer/buffer/prioritized_replay_buffer.py) run on synthetic
#    transitions at a capacity that is NOT a power of two (ring wrap included): trees after the stores, leaves and
#    weights of two sampled batches (the uniform draws recorded), trees after two priority updates (duplicate indices
#    included), leaves of a third batch
# ------------------------------------------------------------------------------------------
def golden_per():
    from gops.trainer.buffer.prioritized_replay_buffer import PrioritizedReplayBuffer as RefPER
    rng = np.random.RandomState(77)
    bkw = dict(trainer="off_serial_trainer", seed=5, obsv_dim=3, action_dim=2, buffer_max_size=37, additional_info={})
    buf = RefPER(index=0, **bkw)
    n = 50
    t = dict(obs=rng.randn(n, 3).astype(np.float32), act=rng.randn(n, 2).astype(np.float32), rew=rng.randn(n).astype(np.float32),
             done=(rng.rand(n) < 0.1).astype(np.float32), obs2=rng.randn(n, 3).astype(np.float32))
    out = {"t/" + k: v for k, v in t.items()}
    draws = []
    orig_uniform = np.random.uniform

    def rec_uniform(low, high=None, size=None):
        v = orig_uniform(low, high, size)
        draws.append((np.asarray(low, dtype=np.float64), np.asarray(high, dtype=np.float64), np.asarray(v, dtype=np.float64)))
        return v

    def store(lo, hi):
        for i in range(lo, hi):
            buf.store(t["obs"][i], t["act"][i], float(t["rew"][i]), bool(t["done"][i]), {}, t["obs2"][i], {}, 0.25)

    def snap(tag):
        out[tag + "/sum_tree"], out[tag + "/min_tree"] = buf.sum_tree.copy(), buf.min_tree.copy()
        out[tag + "/max_priority"], out[tag + "/beta"] = np.float64(buf.max_priority), np.float64(buf.beta)
        out[tag + "/size"], out[tag + "/ptr"] = buf.size, buf.ptr

    store(0, 20)
    snap("s0")
    np.random.seed(3)
    np.random.uniform = rec_uniform
    try:
        for k in range(3):
            if k == 1:
                store(20, 50)           # wraps the ring: 37 slots
                snap("s1")
            b = buf.sample_batch(8)
            lo, hi, v = draws[-1]
            out[f"b{k}/u"] = (v - lo) / (hi - lo)          # the unit draws behind the stratified values
            out[f"b{k}/idx"], out[f"b{k}/weight"] = b["idx"].numpy(), b["weight"].numpy()
            out[f"b{k}/obs"], out[f"b{k}/rew"] = b["obs"].numpy(), b["rew"].numpy()
            pr = np.abs(rng.randn(8)) * (10.0 if k == 1 else 1.0)
            idx = b["idx"].numpy().copy()
            if k == 1:
                idx[5] = idx[2]          # a duplicated index: the later priority wins
            out[f"b{k}/upd_idx"], out[f"b{k}/upd_pr"] = idx, pr
            buf.update_batch(torch.as_tensor(idx), torch.as_tensor(pr))
            snap(f"u{k}")
    finally:
        np.random.uniform = orig_uniform
    out["meta/buffer_kwargs"] = json.dumps(bkw)
    save("per_buffer", **out)


if __name__ == "__main__":
    if sys.argv[1:] == ["per"]:
        golden_per()
        sys.exit(0)
    which = sys.argv[1:] or ["steps", "small", "big", "trained", "fhadp2", "dataenv", "constrained", "penalty", "obsscale", "mac", "spil", "gym", "veh2dof", "errcstr", "mpg", "refpara", "repeat", "nomask", "mobilerobot"]
    if "mobilerobot" in which:
        golden_mobilerobot()
    if "veh2dof" in which:
        golden_steps(VEH2_STEP_CASES)
        golden_small(VEH2_SMALL)
    if "gym" in which:
        golden_steps(GYM_STEP_CASES)
        np.random.seed(0)
        golden_small(GYM_SMALL)
    if "errcstr" in which:
        golden_constrained(ERR_STEP_CASES, ERR_ALG_CASES)
    if "nomask" in which:
        golden_steps(NOMASK_STEP_CASES)
        np.random.seed(0)
        golden_small(NOMASK_SMALL)
    if "repeat" in which:
        golden_steps(REPEAT_STEP_CASES)
        np.random.seed(0)
        golden_small(REPEAT_SMALL)
    if "refpara" in which:
        golden_steps(REFPARA_STEP_CASES)
        golden_small(REFPARA_SMALL)
    if "mpg" in which:
        golden_mpg()
    if "spil" in which:
        golden_spil()
    if "mac" in which:
        np.random.seed(0)   # the reference's (inert) Bayes estimator draws from numpy's global RNG
        golden_small(MAC_SMALL)
    if "obsscale" in which:
        golden_steps(OBS_SCALE_STEP_CASES)
        golden_small(OBS_SCALE_SMALL)
    if "penalty" in which:
        golden_constrained(PENALTY_STEP_CASES, PENALTY_ALG_CASES)
        golden_small(PENALTY_SMALL)
    if "dataenv" in which:
        golden_data_envs()
    if "dataenv_cartpole" in which:
        golden_data_envs(only=("dataenv_cartpole",))
    if "dataenv_veh2dof" in which:
        golden_data_envs(only=("dataenv_veh2dof_p10",))
    if "dataenv_mobilerobot" in which:
        golden_data_envs(only=("dataenv_mobilerobot",))
    if "constrained" in which:
        golden_constrained()
    if "fhadp2" in which:
        golden_fhadp2()
    if "trained" in which:
        golden_trained()
    if "trained256" in which:
        golden_trained256()
    if "steps" in which:
        golden_steps()
    if "small" in which:
        golden_small()
    if "big" in which:
        golden_big()
