"""Closed-loop batched sampling on the GPU.

The reference's samplers step numpy data environments one env and one step at a time on the CPU
(gops/trainer/sampler/base.py:101-187) - the bottleneck once the learner runs on the MI355X (SURVEY
section 8(f) rank 3).  The data environments of this path share their dynamics and stage reward with the env
models but NOT their termination rules or terminal reward (pyth_veh3dofconti.py:224-226,263-271: -100 at done,
world-frame |dx| > 5, |dy| > 2 against the model's ego-frame 10 / 10; lq_base.py:224-239: done when the state leaves
its bounds, -100, no clipping; pyth_idpendulum.py:71-87 is identical to its model; gym_cartpoleconti.py:102-137: reward 1
also for the step that ends an episode; pyth_veh2dofconti.py:179-219: the model's step, -100 at done; pyth_mobilerobot.py:108-152:
the model's step with both headings clipped to +-pi, obstacle noise drawn on the device per step).  `env_step="data"` (default)
selects those data-env semantics in the step kernel (`GopsEnv.data_env`, checked against transitions recorded from the
reference's numpy envs: tests/golden/dataenv_*.npz); `env_step="model"` steps the env model instead.  N environment
instances are advanced together by `gops_env_step`:

    act = policy(obs)            # N x obs_dim through the policy MLP (library GEMMs), on the device
    obs2, rew, done, info2 = gops_env_step(obs, act, done=0, info)
    finished or timed-out instances are re-seeded from a device pool of reset states

`sample()` returns replay-format DEVICE tensors (`obs, act, rew, done, obs2, logp` + info / next_info
keys) for `ReplayBuffer.add_tensors`; nothing crosses PCIe in steady state.  Reset states come from the
data envs' reset distributions (`gops_amd.utils.synthetic.make_batch`), generated on the host in pools of
`pool_factor x n_envs` and uploaded once per pool.
"""
import time

import numpy as np
import torch

from gops_amd import hip_backend as hb
from gops_amd.utils.synthetic import make_batch
from gops_amd.utils.tensorboard_setup import tb_tags

_INFO = ("state", "ref_points", "path_num", "u_num", "ref_time")


class DeviceEnvSampler:
    on_device = True   # trainers: do not move the networks to the CPU around sample()

    def __init__(self, cfg: dict, env_model, *, n_envs: int, steps_per_sample: int = 1, max_episode_steps: int = 200,
                 seed: int = 0, device="cuda", pool_factor: int = 8, noise_std: float = 0.0, env_step: str = "data"):
        """cfg: workload dict as in `gops_amd.utils.synthetic.CONFIGS` (env_id, pre_horizon, lq_config);
        env_model: the wrapped model from `create_env_model` (its constants drive the step kernel)."""
        self.cfg, self.env_model = dict(cfg), env_model
        if getattr(getattr(env_model, "unwrapped", env_model), "ref_c", None) is not None:
            raise RuntimeError("DeviceEnvSampler draws its reset states from the DEFAULT reference trajectories "
                               "(gops_amd.utils.synthetic); a model with custom path_para / u_para needs its own reset pool")
        self.n, self.steps, self.max_steps = n_envs, steps_per_sample, max_episode_steps
        self.device = torch.device(device)
        self.seed, self.pool_factor, self.noise_std = seed, pool_factor, noise_std
        if env_step not in ("data", "model"):
            raise ValueError("env_step must be 'data' (the data environment's step) or 'model' (the env model's)")
        self.data_env = env_step == "data"
        kind = getattr(getattr(env_model, "unwrapped", env_model), "hip_kind", None)
        if self.data_env and (kind not in (hb.ENV_LQ, hb.ENV_IDP, hb.ENV_VEH, hb.ENV_CARTPOLE, hb.ENV_VEH2DOF, hb.ENV_MOBILEROBOT)
                              or self.cfg.get("env_id", "").endswith("errcstr")):
            raise RuntimeError(f"the DATA environment of {self.cfg.get('env_id')} is not restated in the step kernel "
                               "(pyth_lq, pyth_idpendulum, pyth_veh3dofconti, pyth_veh2dofconti, pyth_mobilerobot and "
                               "gym_cartpoleconti are): pass env_step='model'")
        self.networks = None
        self.total = 0
        self._pool, self._pool_pos, self._pools_made = None, 0, 0
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed)
        self._henv = None
        first = self._draw(self.n)
        self.obs = first["obs"]
        self.info = {k: first[k] for k in _INFO if k in first}
        self.t = torch.zeros(self.n, dtype=torch.int32, device=self.device)

    # ---- reset pool ---------------------------------------------------------------------------
    def _draw(self, k: int):
        """k fresh initial conditions (device tensors) from the pool; refills the pool when it runs dry."""
        size = self.pool_factor * self.n
        if self._pool is None or self._pool_pos + k > size:
            host = make_batch(self.cfg, self.seed + 7919 * self._pools_made, batch=size)
            if self.cfg["env_id"] == "pyth_mobilerobot" and self.data_env:
                # the data env's own reset distribution (pyth_mobilerobot.py:31-54, 95-106: robot and obstacle uniform in the work
                # space, w = 0, tracking errors of the robot state); make_batch's near-collision starts exist for the parity fixtures
                rng = np.random.RandomState(self.seed + 7919 * self._pools_made)
                ego = rng.uniform([0.0, -1.0, -0.6, 0.0, 0.0], [2.7, 1.0, 0.6, 0.3, 0.0], size=(size, 5))
                obst = rng.uniform([3.5, -3.0, np.pi / 2 - 0.3, 0.0, 0.0], [6.0, 3.0, np.pi / 2 + 0.3, 0.5, 0.0], size=(size, 5))
                ego, obst = ego.astype(np.float32), obst.astype(np.float32)   # (reset casts the drawn state first, :100-101)
                track = np.stack((ego[:, 1], ego[:, 2], ego[:, 3] - np.float32(0.3)), axis=1)   # path y = 0, phi = 0, v_desired 0.3
                host["obs"] = torch.from_numpy(np.concatenate((ego, track, obst), axis=1))
            if self.cfg["env_id"] == "gym_cartpoleconti" and self.data_env:
                # the data env's own reset distribution (env_gym/gym_cartpoleconti.py:139-147: uniform +-0.05 in every state);
                # make_batch's wide cartpole states exist to exercise the done thresholds inside short rollouts
                rng = np.random.RandomState(self.seed + 7919 * self._pools_made)
                host["obs"] = torch.from_numpy(rng.uniform(-0.05, 0.05, size=(size, 4)).astype(np.float32))
            self._pool = {key: v.to(self.device) for key, v in host.items() if key == "obs" or key in _INFO}
            # ScaleObservationData (scale_observation.py:53-62): the sampler hands out (obs + shift) * scale
            sc, sh = getattr(self.env_model, "obs_scale", None), getattr(self.env_model, "obs_shift", None)
            if sc is not None or sh is not None:
                as_t = lambda v, d: torch.as_tensor(d if v is None else v, dtype=torch.float32, device=self.device)  # noqa: E731
                self._pool["obs"] = (self._pool["obs"] + as_t(sh, 0.0)) * as_t(sc, 1.0)
            self._pool_pos, self._pools_made = 0, self._pools_made + 1
        sl = slice(self._pool_pos, self._pool_pos + k)
        self._pool_pos += k
        return {key: v[sl].clone() for key, v in self._pool.items()}

    def _hip_env(self):
        if self._henv is None:
            pol = self.networks.policy
            self._henv = self.env_model.hip_env(pol.act_low_lim.cpu().numpy(), pol.act_high_lim.cpu().numpy(),
                                                data_env=self.data_env)
        return self._henv

    # ---- sampling -----------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self):
        """Advance all N environments `steps_per_sample` steps; returns (dict of [N*steps, ...] device
        tensors in replay format, tb dict)."""
        t0 = time.time()
        policy = self.networks.policy
        zeros = torch.zeros(self.n, device=self.device)
        chunks = []
        for _ in range(self.steps):
            obs, info = self.obs, self.info
            act = policy(obs)
            if self.noise_std > 0.0:
                act = act + self.noise_std * torch.randn(act.shape, generator=self._gen, device=self.device)
            obs2, rew, done, info2 = hb.env_step(self._hip_env(), obs, act.contiguous(), zeros, info)
            row = dict(obs=obs, act=act, rew=rew, done=done, obs2=obs2, logp=zeros)
            for k in info:
                row[k], row["next_" + k] = info[k], info2[k]
            chunks.append(row)
            # episode bookkeeping: terminated or timed-out instances restart from a fresh reset state
            self.t += 1
            over = (done != 0) | (self.t >= self.max_steps)
            n_over = int(over.sum().item())
            # (pyth_mobilerobot's info["constraint"] is a zero-size replay slot in the reference, pyth_mobilerobot.py:86-92:
            # the step's constraint output is not a stored column)
            self.obs, self.info = obs2, {k: info2[k] for k in info}
            if n_over:
                fresh = self._draw(n_over)
                idx = over.nonzero(as_tuple=True)[0]
                self.obs = self.obs.index_copy(0, idx, fresh["obs"])
                for k in self.info:
                    self.info[k] = self.info[k].index_copy(0, idx, fresh[k])
                self.t.index_fill_(0, idx, 0)
        batch = {k: torch.cat([c[k] for c in chunks]) for k in chunks[0]} if len(chunks) > 1 else chunks[0]
        self.total += self.n * self.steps
        return batch, {tb_tags["sampler_time"]: (time.time() - t0) * 1000}

    def sample_with_replay_format(self):
        return self.sample()

    def get_total_sample_number(self):
        return self.total

    def load_state_dict(self, state_dict):
        pass
