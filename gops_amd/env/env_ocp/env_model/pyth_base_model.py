"""Base class of the model-type environments (contract of the reference's
gops/env/env_ocp/env_model/pyth_base_model.py:21-85): dimensions, dt, bound tensors on `device`,
`forward(obs, action, done, info) -> (next_obs, reward, next_done, next_info)`, `.unwrapped`.

Every concrete model here is evaluated on the MI355X: `forward` is one call of `gops_env_step`
(include/gops_hip.h) and the fused horizon rollout reads the same constants through
`hip_constants()`.
"""
from typing import Callable, Dict, Optional, Sequence, Tuple, Union

import torch


class PythBaseModel:
    hip_kind: int = 0  # GOPS_ENV_* id of the HIP implementation

    def __init__(self, obs_dim: int, action_dim: int, dt: Optional[float] = None,
                 obs_lower_bound: Optional[Sequence] = None, obs_upper_bound: Optional[Sequence] = None,
                 action_lower_bound: Optional[Sequence] = None, action_upper_bound: Optional[Sequence] = None,
                 device: Union[torch.device, str, None] = None):
        self.obs_dim, self.action_dim, self.dt, self.device = obs_dim, action_dim, dt, device

        def bound(v, n, fill):
            v = [fill] * n if v is None else v
            return torch.tensor(v, dtype=torch.float32, device=device)

        self.obs_lower_bound = bound(obs_lower_bound, obs_dim, float("-inf"))
        self.obs_upper_bound = bound(obs_upper_bound, obs_dim, float("inf"))
        self.action_lower_bound = bound(action_lower_bound, action_dim, float("-inf"))
        self.action_upper_bound = bound(action_upper_bound, action_dim, float("inf"))

    # optional hooks with the reference's names (pyth_base_model.py:69-81)
    get_constraint: Callable = None
    get_terminal_cost: Callable = None

    def hip_constants(self) -> Dict:
        """Extra keyword arguments for `hip_backend.make_env` (e.g. the LQ matrices)."""
        return {}

    def _hip_get_constraint(self, obs: torch.Tensor, info: Optional[Dict] = None) -> torch.Tensor:
        """`get_constraint(obs, info)` of the models that define one (pyth_base_model.py:69-75): [B, n_constraint], each
        entry required <= 0 - one launch of `gops_env_constraint`."""
        from gops_amd import hip_backend as hb
        if not obs.is_cuda:
            raise RuntimeError("get_constraint runs on the MI355X only (tensors must be on 'cuda'); there is no CPU path in gops_amd")
        env = self.__dict__.get("_constraint_env")
        if env is None:
            env = self._constraint_env = hb.make_env(
                self.hip_kind, self.obs_dim, self.action_dim, act_low=self.action_lower_bound.cpu(),
                act_high=self.action_upper_bound.cpu(), pre_horizon=getattr(self, "pre_horizon", 0), **self.hip_constants())
        f = lambda t: t.to(dtype=torch.float32).contiguous()
        dev_info = {k: f(info[k]) for k in ("state", "surr_state") if info and info.get(k) is not None}
        return hb.env_constraint(env, f(obs), dev_info)

    def forward(self, obs: torch.Tensor, action: torch.Tensor, done: torch.Tensor, info: Dict
                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, Dict]:
        raise NotImplementedError

    @property
    def unwrapped(self):
        return self
