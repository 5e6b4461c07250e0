"""create_env_model: string-keyed registry of model-type environments plus the wrapper chain.

Same surface as the reference's gops/create_pkg/create_env_model.py:51-147 (registry filled by
scanning `env/env_*/env_model/*.py` for `env_model_creator` or the CamelCase class; the same
keyword arguments select the wrappers).  Where the reference nests one Python object per wrapper
(ScaleAction -> ClipAction -> ClipObservation -> ShapingReward -> MaskAtDone -> base), here the
chain is a set of constants on ONE object, because the whole chain is evaluated inside the HIP
kernels (csrc/common.h wrap_action, csrc/rollout_fwd.hip).
"""
import os
from typing import Callable, Dict, Optional, Tuple, Union

import numpy as np
import torch

from gops_amd import hip_backend as hb
from gops_amd.create_pkg._registry import Registry
from gops_amd.utils.gops_path import env_path, underline2camel

registry = Registry("env")


def register(env_id: str, entry_point: Union[Callable, str], **kwargs):
    registry.add(env_id, entry_point, kwargs, env_id=env_id)


class WrappedEnvModel:
    """The wrapped model: base model + ScaleAction/ClipAction/ClipObservation/ShapingReward/
    MaskAtDone constants.  `forward` is one fused `gops_env_step`; `hip_env()` exports the
    constants for the fused horizon rollout."""

    def __init__(self, model, *, min_action, max_action, clip_obs: bool,
                 reward_scale: Optional[float], reward_shift: Optional[float], obs_scale=None, obs_shift=None,
                 repeat_num: Optional[int] = None, sum_reward: bool = True, mask_at_done: bool = True,
                 strict_reference_points: bool = False):
        self.model = model
        # opt-in (models with reference trajectories): `forward` takes the appended reference point from the host's torch CPU ops -
        # the reference's own fp32 values on this host - instead of the kernel's evaluation (env/env_ocp/resources/ref_traj_host.py)
        self.strict_reference_points = bool(strict_reference_points) and model.hip_kind in (hb.ENV_VEH, hb.ENV_VEH_SURR, hb.ENV_VEH2DOF)
        self._host_traj = None
        self.mask_at_done = mask_at_done   # False: no MaskAtDoneModel in the chain (GopsEnv.no_mask_at_done)
        self.repeat_num, self.sum_reward = repeat_num, sum_reward   # ActionRepeatModel constants (None: no such wrapper)
        self.obs_scale, self.obs_shift = obs_scale, obs_shift   # ScaleObservationModel constants (None: no such wrapper)
        self.min_action = torch.zeros_like(model.action_lower_bound) + torch.as_tensor(
            min_action, dtype=torch.float32, device=model.action_lower_bound.device)
        self.max_action = torch.zeros_like(model.action_upper_bound) + torch.as_tensor(
            max_action, dtype=torch.float32, device=model.action_upper_bound.device)
        # like ScaleActionModel, the wrapped model advertises the scaled action range
        self.action_lower_bound, self.action_upper_bound = self.min_action, self.max_action
        self.clip_obs = clip_obs
        self.reward_scale, self.reward_shift = reward_scale, reward_shift
        self._env_cache = {}

    def __getattr__(self, name):   # obs_dim, dt, obs bounds, get_constraint, ... from the base model
        return getattr(self.__dict__["model"], name)

    @property
    def unwrapped(self):
        return self.model.unwrapped

    def hip_env(self, policy_low=None, policy_high=None, data_env: bool = False) -> hb.GopsEnv:
        """C-ABI constants of the wrapped model.  `data_env=True`: the variant whose `gops_env_step` reproduces the DATA
        environment's step (termination tests, -100 terminal penalty; `sampler/device_env_sampler.py`)."""
        key = (None if policy_low is None else tuple(np.asarray(policy_low, dtype=np.float32).reshape(-1).tolist()),
               None if policy_high is None else tuple(np.asarray(policy_high, dtype=np.float32).reshape(-1).tolist()),
               bool(data_env))
        if key not in self._env_cache:
            m = self.model
            self._env_cache[key] = hb.make_env(
                m.hip_kind, m.obs_dim, m.action_dim, act_low=m.action_lower_bound.cpu(),
                act_high=m.action_upper_bound.cpu(), min_action=self.min_action.cpu(),
                max_action=self.max_action.cpu(), policy_low=policy_low, policy_high=policy_high,
                obs_low=m.obs_lower_bound.cpu() if (self.clip_obs or data_env) else None,
                obs_high=m.obs_upper_bound.cpu() if (self.clip_obs or data_env) else None,
                pre_horizon=getattr(m, "pre_horizon", 0), reward_scale=self.reward_scale,
                reward_shift=self.reward_shift, data_env=data_env, obs_scale=self.obs_scale, obs_shift=self.obs_shift,
                ref_c=getattr(m, "ref_c", None), repeat_num=self.repeat_num, sum_reward=self.sum_reward,
                mask_at_done=self.mask_at_done, **m.hip_constants())
        return self._env_cache[key]

    def forward(self, obs: torch.Tensor, action: torch.Tensor, done: torch.Tensor, info: Dict
                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, Dict]:
        if not obs.is_cuda:
            raise RuntimeError("env_model.forward runs on the MI355X only (tensors must be on 'cuda'); "
                               "there is no CPU fallback in gops_amd")
        f = lambda t: t.to(torch.float32).contiguous()  # noqa: E731
        dev_info = {k: f(info[k]) for k in ("state", "ref_points", "path_num", "u_num", "ref_time", "surr_state", "noise", "ref_appended")
                    if isinstance(info, dict) and k in info and info[k] is not None}
        if self.strict_reference_points and "ref_appended" not in dev_info:
            # the point this step appends, at (ref_time + dt) + pre_horizon * dt (pyth_veh3dofconti_model.py:106-128), from the host
            if self._host_traj is None:
                from gops_amd.env.env_ocp.resources.ref_traj_host import HostRefTraj
                self._host_traj = HostRefTraj(getattr(self.model, "ref_c", None), dt=self.model.dt)
            pts = self._host_traj.appended_points(dev_info["ref_time"], dev_info["path_num"], dev_info["u_num"], 1, self.model.pre_horizon)
            dev_info["ref_appended"] = pts[:, 0].contiguous().to(obs.device)
        nobs, rew, ndone, ninfo = hb.env_step(self.hip_env(), f(obs), f(action), f(done), dev_info)
        if "constraint" not in ninfo:
            ninfo["constraint"] = None
        return nobs, rew, ndone.bool(), ninfo


def create_env_model(
    env_id: str,
    *,
    reward_shift: Optional[float] = None,
    reward_scale: Optional[float] = None,
    obs_shift: Union[np.ndarray, float, list, None] = None,
    obs_scale: Union[np.ndarray, float, list, None] = None,
    clip_obs: bool = True,
    clip_action: bool = True,
    mask_at_done: bool = True,
    repeat_num: Optional[int] = None,
    sum_reward: bool = True,
    action_scale: bool = True,
    min_action: Union[float, int, np.ndarray, list] = -1.0,
    max_action: Union[float, int, np.ndarray, list] = 1.0,
    strict_reference_points: bool = False,
    **kwargs,
) -> object:
    """Build the model `<env_id>_model` and apply the wrappers selected by the arguments (same
    arguments, defaults and KeyError/RuntimeError behaviour as the reference)."""
    env_model = registry.build(env_id + "_model", **kwargs, device="cuda" if kwargs.get("use_gpu", False) else "cpu")

    # wrapper options outside the fused kernels' contract are refused, never silently ignored
    if repeat_num is not None:   # ActionRepeatModel: in the kernels of the models whose observation is the state
        if env_model.hip_kind not in (hb.ENV_LQ, hb.ENV_IDP, hb.ENV_CARTPOLE, hb.ENV_PENDULUM):
            raise RuntimeError("ActionRepeatModel (repeat_num) is supported for pyth_lq / pyth_idpendulum / gym_* models only "
                               "by the HIP env models (the reference wrapper does not advance `info`)")
        if not 1 <= int(repeat_num) <= hb.MAX_REPEAT:
            raise RuntimeError(f"repeat_num must be 1..{hb.MAX_REPEAT} for the HIP env models")
    scaled = obs_shift is not None or obs_scale is not None
    if scaled and (env_model.obs_dim > 8 or env_model.hip_kind not in (hb.ENV_LQ, hb.ENV_IDP, hb.ENV_CARTPOLE, hb.ENV_PENDULUM)):
        raise RuntimeError("ScaleObservationModel (obs_shift/obs_scale) is supported for pyth_lq / pyth_idpendulum / gym_* models only "
                           "(observation dimension <= 8) by the HIP env models")
    if not mask_at_done and env_model.hip_kind == hb.ENV_VEH_SURR:
        raise RuntimeError("mask_at_done=False is not supported for the constrained veh3dofconti models by the HIP env models")
    # action_scale=True with clip_action=False needs no flag: ScaleActionModel already ends with
    # clip(., action_lower_bound, action_upper_bound) (scale_action.py:75-83) and ClipActionModel applies that same
    # clamp once more (clip_action.py:34-36) - idempotent, so the kernels' chain is exact for both settings.
    if not action_scale:
        if not clip_action:
            raise RuntimeError("action_scale=False with clip_action=False is not supported by the HIP env models")
        # ClipActionModel alone == scaling from [low, high] onto itself
        min_action, max_action = env_model.action_lower_bound.cpu().numpy(), env_model.action_upper_bound.cpu().numpy()
    shaping = reward_scale is not None or reward_shift is not None
    return WrappedEnvModel(
        env_model, min_action=min_action, max_action=max_action, clip_obs=clip_obs,
        reward_scale=(1.0 if reward_scale is None else reward_scale) if shaping else None,
        reward_shift=(0.0 if reward_shift is None else reward_shift) if shaping else None,
        # create_env_model.py:115-118: either one given -> the wrapper is applied with the other at its neutral value
        obs_scale=(1.0 if obs_scale is None else obs_scale) if scaled else None,
        obs_shift=(0.0 if obs_shift is None else obs_shift) if scaled else None,
        repeat_num=None if repeat_num is None else int(repeat_num), sum_reward=bool(sum_reward), mask_at_done=bool(mask_at_done),
        strict_reference_points=strict_reference_points)


# fill the registry: every env/env_*/env_model/<id>.py exporting env_model_creator or the CamelCase class
def _model_entries(stem, module):
    ctor = getattr(module, "env_model_creator", None) or getattr(module, underline2camel(stem), None)
    if ctor is None:
        print(f"env {stem} has no env_model_creator or {underline2camel(stem)}")
        return
    yield stem, ctor, dict(env_id=stem)


for _env_dir in sorted(e for e in os.listdir(env_path) if e.startswith("env_")):
    _model_dir = os.path.join(env_path, _env_dir, "env_model")
    if os.path.isdir(_model_dir):
        registry.scan(_model_dir, f"gops_amd.env.{_env_dir}.env_model", _model_entries,
                      keep=lambda stem: "base" not in stem)
