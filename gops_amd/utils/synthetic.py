"""Seeded synthetic batches at the BASELINE.json shapes.

Host-side input generation only (numpy); mirrors what the reference's *data* environments put
into a replay batch: `pyth_idpendulum.py:36-38` (uniform initial state), `pyth_veh3dofconti.py:
91-193` (reference time / path / speed ids, P+1 reference points evaluated in float64 and cast
to float32, state = ref_0 + delta), `lq_base.py:150-155` + `pyth_base_env.py:61-65` (uniform in
mean +- 3 sigma).  Used by `bench.py`, the tests and `tests/golden/make_golden.py` so that the
HIP path, the oracle and the reference all see bit-identical inputs.
"""
from typing import Dict

import numpy as np
import torch

# name -> description of every workload in BASELINE.json `configs` (+ the north_star target)
CONFIGS: Dict[str, Dict] = {
    # configs[0]: plumbing case the reference runs on CPU
    "cfg1_idp_fhadp_b64_h10": dict(alg="FHADP", env_id="pyth_idpendulum", batch=64, horizon=10,
                                   hidden=(64, 64), act="gelu", gamma=1.0),
    # configs[1]
    "cfg2_idp_fhadp_b4096_h30": dict(alg="FHADP", env_id="pyth_idpendulum", batch=4096, horizon=30,
                                     hidden=(256, 256), act="gelu", gamma=1.0),
    # north_star target (metric is quoted on this one)
    "target_veh3dof_fhadp_b4096_h30": dict(alg="FHADP", env_id="pyth_veh3dofconti", batch=4096,
                                           horizon=30, pre_horizon=30, hidden=(256, 256), act="elu",
                                           gamma=1.0),
    # configs[2]
    "cfg3_veh3dof_infadp_b8192": dict(alg="INFADP", env_id="pyth_veh3dofconti", batch=8192,
                                      horizon=10, pre_horizon=10, hidden=(256, 256, 256), act="relu",
                                      gamma=0.99),
    # configs[3] (per replica)
    "cfg4_veh3dof_fhadp_b4096_h50": dict(alg="FHADP", env_id="pyth_veh3dofconti", batch=4096,
                                         horizon=50, pre_horizon=50, hidden=(256, 256), act="elu",
                                         gamma=1.0),
    # configs[4] (per replica)
    "cfg5_lq_infadp_b65536": dict(alg="INFADP", env_id="pyth_lq", lq_config="s4a2", batch=65536,
                                  horizon=10, hidden=(256, 256), act="gelu", gamma=0.99),
}

_LQ_INIT = {  # lq_configs.py init_mean / init_std
    "s2a1": ([0.0, 0.0], [1.0, 1.0]),
    "s3a1": ([0, 0, 0], [2, 2, 2]),
    "s4a2": ([0, 0, 0, 0], [0.7, 0.3, 0.7, 0.3]),
    "s5a1": ([0] * 5, [0.1] * 5),
    "s6a3": ([0] * 6, [0.1] * 6),
}
_LQ_ACT_DIM = {"s2a1": 1, "s3a1": 1, "s4a2": 2, "s5a1": 1, "s6a3": 3}

_W = 2 * np.pi / 10


def ref_points_f64(t: np.ndarray, path_num: np.ndarray, u_num: np.ndarray) -> np.ndarray:
    """(x, y, phi, u)(t) of the data-env reference generator, float64, vectorised.

    Same closed forms as `ref_traj_data.py` (`MultiRefTrajData`): 4 paths x 2 speed profiles,
    heading from a 1 ms forward difference.  t may have any shape; ids broadcast against it.
    """
    t = np.asarray(t, dtype=np.float64)
    path_num = np.broadcast_to(np.asarray(path_num), t.shape)
    u_num = np.broadcast_to(np.asarray(u_num), t.shape)

    def arc(tt):
        return np.where(u_num == 0, -1.0 / _W * np.cos(_W * tt) + 5.0 * tt + 1.0 / _W, 5.0 * tt)

    def xy(tt):
        s = arc(tt)
        y_sine = 1.5 * np.sin(_W * tt)
        y_lane = np.select(
            [tt <= 5.0, tt <= 9.0, tt <= 14.0, tt <= 18.0],
            [0.0, 3.5 / 4.0 * (tt - 5.0), 3.5, -3.5 / 4.0 * (tt - 14.0) + 3.5], 0.0)
        sm = np.mod(tt, 10.0)
        y_tri = np.where(sm <= 5.0, 0.6 * sm, -0.6 * (sm - 10.0))
        x = np.where(path_num == 3, 100.0 * np.sin(s / 100.0), s)
        y = np.select([path_num == 0, path_num == 1, path_num == 2],
                      [y_sine, y_lane, y_tri], 100.0 * (np.cos(s / 100.0) - 1.0))
        return x, y

    x0, y0 = xy(t)
    x1, y1 = xy(t + 0.001)
    phi = np.arctan2(y1 - y0, x1 - x0)
    u = np.where(u_num == 0, np.sin(_W * t) + 5.0, 5.0)
    return np.stack((x0, y0, phi, u), axis=-1)


def _angle_normalize(x):
    return ((x + np.pi) % (2 * np.pi)) - np.pi


def veh_obs_f32(state: np.ndarray, ref_points: np.ndarray) -> np.ndarray:
    """Ego-frame observation from (state [B,6], ref_points [B,P+1,4]) - data-env get_obs."""
    ex, ey, ephi = state[:, 0:1], state[:, 1:2], state[:, 2:3]
    c, s = np.cos(-ephi), np.sin(-ephi)
    dx, dy = ref_points[..., 0] - ex, ref_points[..., 1] - ey
    x_tf = dx * c - dy * s
    y_tf = dx * s + dy * c
    phi_tf = _angle_normalize(ref_points[..., 2] - ephi)
    u_tf = ref_points[..., 3] - state[:, 3:4]
    per_pt = np.stack((x_tf, y_tf, phi_tf, u_tf), axis=2)  # [B,P+1,4]
    ego = np.concatenate((per_pt[:, 0], state[:, 4:6]), axis=1)
    return np.concatenate((ego, per_pt[:, 1:].reshape(state.shape[0], -1)), axis=1).astype(np.float32)


def obs_dim_of(cfg: Dict) -> int:
    if cfg["env_id"] == "pyth_idpendulum":
        return 6
    if cfg["env_id"] == "gym_cartpoleconti":
        return 4
    if cfg["env_id"] in ("pyth_veh2dofconti", "pyth_veh2dofconti_errcstr"):
        return 4 + cfg["pre_horizon"]
    if cfg["env_id"] == "gym_pendulum":
        return 3
    if cfg["env_id"] == "pyth_mobilerobot":
        return 13
    if cfg["env_id"] in ("pyth_veh3dofconti", "pyth_veh3dofconti_errcstr"):
        return 6 + 4 * cfg["pre_horizon"]
    if cfg["env_id"] in _SURR_ENVS:
        return 6 + 4 * cfg["pre_horizon"] + 4 * n_surr_of(cfg)
    return len(_LQ_INIT[cfg.get("lq_config", "s4a2")][0])


_SURR_ENVS = {"pyth_veh3dofconti_surrcstr": 4, "pyth_veh3dofconti_detour": 1, "pyth_veh3dofconti_surrcstr_penalty": 1}   # env id -> default surr_veh_num


def n_surr_of(cfg: Dict) -> int:
    return cfg.get("surr_veh_num", _SURR_ENVS[cfg["env_id"]])


def act_dim_of(cfg: Dict) -> int:
    if cfg["env_id"] in ("pyth_idpendulum", "gym_cartpoleconti", "gym_pendulum", "pyth_veh2dofconti", "pyth_veh2dofconti_errcstr"):
        return 1
    if cfg["env_id"] in ("pyth_veh3dofconti", "pyth_veh3dofconti_errcstr", "pyth_mobilerobot") or cfg["env_id"] in _SURR_ENVS:
        return 2
    return _LQ_ACT_DIM[cfg.get("lq_config", "s4a2")]


def make_batch(cfg: Dict, seed: int, batch: int = None) -> Dict[str, torch.Tensor]:
    """Replay-format batch (all float32, like `replay_buffer.py:100-108`) on CPU."""
    B = cfg["batch"] if batch is None else batch
    rng = np.random.RandomState(seed)
    env_id = cfg["env_id"]
    out: Dict[str, np.ndarray] = {}
    if env_id == "pyth_idpendulum":
        h = np.array([5, 0.1, 0.1, 0.3, 0.3, 0.3], dtype=np.float32)
        out["obs"] = rng.uniform(-h, h, size=(B, 6)).astype(np.float32)
    elif env_id in ("pyth_veh2dofconti", "pyth_veh2dofconti_errcstr"):
        # data env reset (pyth_veh2dofconti.py:121-170): state = (y, phi) of the first reference point + U(+-[1, pi/6]), (v, omega) ~ U(+-0.2)
        P = cfg["pre_horizon"]
        t0 = 20.0 * rng.uniform(0.0, 1.0, size=B)
        path_num = rng.randint(0, 4, size=B)
        u_num = rng.randint(0, 2, size=B)
        tt = t0[:, None] + 0.1 * np.arange(P + 1)[None, :]
        ref4 = ref_points_f64(tt, path_num[:, None], u_num[:, None])
        ref = ref4[:, :, 1:3].astype(np.float32)                         # (y, phi)
        hi = np.array([1.0, np.pi / 6, 0.2, 0.2], dtype=np.float32)
        delta = rng.uniform(-hi, hi, size=(B, 4)).astype(np.float32)
        state = np.concatenate((ref[:, 0, :] + delta[:, :2], delta[:, 2:]), axis=1).astype(np.float32)
        obs = np.concatenate((state[:, :2] - ref[:, 0], state[:, 2:], state[:, :1] - ref[:, 1:, 0]), axis=1).astype(np.float32)
        out.update(obs=obs, state=state, ref_points=ref, path_num=path_num.astype(np.float32),
                   u_num=u_num.astype(np.float32), ref_time=t0.astype(np.float32))
    elif env_id == "pyth_mobilerobot":
        # data env reset (pyth_mobilerobot.py:31-54,98-109): ego (x, y, theta, v, w) and obstacle uniform in the work space,
        # tracking errors of the ego state; here the ego also turns (w != 0) and a third of the obstacles start within
        # 0.6 .. 1.2 m of the ego, so that the safety constraint (0.89 m) and the collision test (0.74 m) are exercised
        ego = rng.uniform([0.0, -1.0, -0.6, 0.0, -0.5], [2.7, 1.0, 0.6, 0.4, 0.5], size=(B, 5))
        obst = rng.uniform([3.5, -3.0, np.pi / 2 - 0.3, 0.0, -0.3], [6.0, 3.0, np.pi / 2 + 0.3, 0.5, 0.3], size=(B, 5))
        near = rng.uniform(size=B) < 0.35
        ang, rad = rng.uniform(-np.pi, np.pi, size=B), rng.uniform(0.6, 1.2, size=B)
        obst[:, 0] = np.where(near, ego[:, 0] + rad * np.cos(ang), obst[:, 0])
        obst[:, 1] = np.where(near, ego[:, 1] + rad * np.sin(ang), obst[:, 1])
        track = np.stack((ego[:, 1], ego[:, 2], ego[:, 3] - 0.3), axis=1)
        out["obs"] = np.concatenate((ego, track, obst), axis=1).astype(np.float32)
    elif env_id == "gym_cartpoleconti":   # wide enough that some trajectories leave |x| <= 2.4 / |theta| <= 12 deg within a rollout
        h = np.array([2.3, 1.0, 0.2, 1.0], dtype=np.float32)
        out["obs"] = rng.uniform(-h, h, size=(B, 4)).astype(np.float32)
    elif env_id == "gym_pendulum":        # (cos th, sin th, thdot), speeds up to the +-8 clamp
        th = rng.uniform(-np.pi, np.pi, size=B)
        out["obs"] = np.stack((np.cos(th), np.sin(th), rng.uniform(-8.0, 8.0, size=B)), axis=1).astype(np.float32)
    elif env_id == "pyth_lq":
        mean, std = (np.array(v, dtype=np.float32) for v in _LQ_INIT[cfg.get("lq_config", "s4a2")])
        out["obs"] = rng.uniform(mean - 3 * std, mean + 3 * std, size=(B, len(mean))).astype(np.float32)
    elif env_id in ("pyth_veh3dofconti", "pyth_veh3dofconti_errcstr") or env_id in _SURR_ENVS:
        P = cfg["pre_horizon"]
        t0 = 20.0 * rng.uniform(0.0, 1.0, size=B)
        path_num = rng.randint(0, 4, size=B)
        u_num = rng.randint(0, 2, size=B)
        tt = t0[:, None] + 0.1 * np.arange(P + 1)[None, :]
        ref = ref_points_f64(tt, path_num[:, None], u_num[:, None]).astype(np.float32)
        hi = np.array([2, 1, np.pi / 6, 2, 0.1, 0.1], dtype=np.float32)
        delta = rng.uniform(-hi, hi, size=(B, 6)).astype(np.float32)
        state = np.concatenate((ref[:, 0, :4] + delta[:, :4], delta[:, 4:]), axis=1).astype(np.float32)
        out.update(obs=veh_obs_f32(state, ref), state=state, ref_points=ref,
                   path_num=path_num.astype(np.float32), u_num=u_num.astype(np.float32),
                   ref_time=t0.astype(np.float32))
        if env_id in _SURR_ENVS:
            # surrounding vehicles around the first reference point (data env reset, pyth_veh3dofconti_surrcstr.py:81-116:
            # longitudinal / lateral offsets, heading of the road, speed 5 +- 1); a share of them close enough to the ego
            # vehicle that the collision constraint is active, small steering angles so that tan(delta) is exercised
            ns = n_surr_of(cfg)
            lon = rng.uniform(3.0, 14.0, size=(B, ns)) * rng.choice([-1.0, 1.0], size=(B, ns))
            lat = rng.uniform(-3.5, 3.5, size=(B, ns))
            sphi = ref[:, :1, 2] * (path_num[:, None] == 3) + rng.uniform(-0.05, 0.05, size=(B, ns))
            sx = ref[:, :1, 0] + lon * np.cos(sphi) - lat * np.sin(sphi)
            sy = ref[:, :1, 1] + lon * np.sin(sphi) + lat * np.cos(sphi)
            su = 5.0 + rng.uniform(-1.0, 1.0, size=(B, ns))
            if env_id.endswith("detour"):
                su = su * (rng.uniform(size=(B, ns)) < 0.5)   # the detour env parks its vehicle (u = 0): keep half of them parked
            sdelta = rng.uniform(-0.03, 0.03, size=(B, ns))
            surr = np.stack((sx, sy, sphi, su, sdelta), axis=2).astype(np.float32)
            rel = (surr[..., :4] - state[:, None, :4]).reshape(B, -1)
            if env_id.endswith("penalty"):   # the penalty model observes its vehicle in the ego frame (:117-125)
                dx, dy = surr[:, 0, 0] - state[:, 0], surr[:, 0, 1] - state[:, 1]
                c, sn = np.cos(-state[:, 2]), np.sin(-state[:, 2])
                dphi = surr[:, 0, 2] - state[:, 2]
                rel = np.stack((dx * c - dy * sn, dx * sn + dy * c, (dphi + np.pi) % (2 * np.pi) - np.pi,
                                surr[:, 0, 3] - state[:, 3]), axis=1)
            out["obs"] = np.concatenate((out["obs"], rel), axis=1).astype(np.float32)
            out["surr_state"] = surr
    else:
        raise KeyError(env_id)
    A = act_dim_of(cfg)
    out["done"] = np.zeros(B, dtype=np.float32)
    out["act"] = np.zeros((B, A), dtype=np.float32)
    out["rew"] = np.zeros(B, dtype=np.float32)
    out["obs2"] = out["obs"].copy()
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in out.items()}
