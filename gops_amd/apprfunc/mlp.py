"""MLP approximate functions of the ADP path: DetermPolicy, FiniteHorizonPolicy,
FiniteHorizonFullPolicy, StateValue, ActionValue.

Module structure, parameter names (`pi.0.weight` ... / `v.0.weight` ...), registered buffers and
`forward` semantics follow the reference (gops/apprfunc/mlp.py:36-41,50-111,309-329) so that its
`apprfunc_*.pkl` checkpoints load unchanged.  `forward` is the eager definition used by samplers
and evaluators for single observations; inside `compute_gradient` the same parameters are read
in place by the fused HIP rollout (`hip_mlp()`), which never calls `forward`.
"""
__all__ = ["DetermPolicy", "FiniteHorizonPolicy", "FiniteHorizonFullPolicy", "StateValue", "ActionValue"]

import weakref

import torch
import torch.nn as nn

from gops_amd.utils.act_distribution import Action_Distribution
from gops_amd.utils.common_utils import get_activation_func


def mlp(sizes, activation, output_activation=nn.Identity):
    layers = []
    for j in range(len(sizes) - 1):
        act = activation if j < len(sizes) - 2 else output_activation
        layers += [nn.Linear(sizes[j], sizes[j + 1]), act()]
    return nn.Sequential(*layers)


# ctypes structs (raw device pointers) cannot be pickled / deep-copied, and trainers deep-copy the
# networks for host-side samplers: the C-ABI views live OUTSIDE the modules, keyed weakly by module.
_HIP_CACHE = weakref.WeakKeyDictionary()


class _HipMlpMixin:
    """Exports the Linear stack as a C-ABI `GopsMlp` over the live parameter storage."""

    _net_attr = "pi"

    def linear_layers(self):
        layers = getattr(self, "_linear_cache", None)
        if layers is None:
            layers = [m for m in getattr(self, self._net_attr) if isinstance(m, nn.Linear)]
            object.__setattr__(self, "_linear_cache", layers)
        return layers

    def hip_mlp(self, dtype=None, variant_flags=None):
        """`variant_flags` (GOPS_VF_*, None = hip_backend.DEFAULT_VARIANT_FLAGS) only matter for plain MLP batches (`ValueNet`,
        `MlpNet`); a rollout takes its own."""
        from gops_amd import hip_backend as hb
        if self._output_activation != "linear":
            raise RuntimeError("the HIP rollout supports a linear output activation only")
        layers = self.linear_layers()
        # the struct only holds raw pointers: rebuild it when the storage moved (.to(device), load)
        key = tuple(l.weight.data_ptr() for l in layers) + tuple(l.bias.data_ptr() for l in layers) + (hb.dtype_id(dtype), variant_flags)
        cached = _HIP_CACHE.get(self)
        if cached is None or cached[0] != key:
            mlp = hb.make_mlp([l.weight.data for l in layers], [l.bias.data for l in layers],
                              self._hidden_activation, dtype, variant_flags=variant_flags)
            _HIP_CACHE[self] = (key, mlp)
            return mlp
        return cached[1]


class DetermPolicy(nn.Module, Action_Distribution, _HipMlpMixin):
    """Deterministic policy: obs -> tanh-squashed action."""

    _time_input = False

    def __init__(self, **kwargs):
        super().__init__()
        obs_dim = kwargs["obs_dim"] + (1 if self._time_input else 0)
        sizes = [obs_dim] + list(kwargs["hidden_sizes"]) + [kwargs["act_dim"]]
        self._hidden_activation = kwargs["hidden_activation"]
        self._output_activation = kwargs.get("output_activation", "linear")
        self.pi = mlp(sizes, get_activation_func(self._hidden_activation),
                      get_activation_func(self._output_activation))
        self.register_buffer("act_high_lim", torch.from_numpy(kwargs["act_high_lim"]))
        self.register_buffer("act_low_lim", torch.from_numpy(kwargs["act_low_lim"]))
        self.action_distribution_cls = kwargs["action_distribution_cls"]

    def _squash(self, y):
        return (self.act_high_lim - self.act_low_lim) / 2 * torch.tanh(y) + (self.act_high_lim + self.act_low_lim) / 2

    def forward(self, obs):
        return self._squash(self.pi(obs))


class FiniteHorizonPolicy(DetermPolicy):
    """Finite-horizon policy: the (un-normalised) virtual time step is one more input column."""

    _time_input = True

    def forward(self, obs, virtual_t=1):
        t = virtual_t * torch.ones(size=[obs.shape[0], 1], dtype=torch.float32, device=obs.device)
        return self._squash(self.pi(torch.cat((obs, t), 1)))


class FiniteHorizonFullPolicy(nn.Module, Action_Distribution, _HipMlpMixin):
    """Finite-horizon policy that emits the whole action sequence from one evaluation at obs_0
    (reference gops/apprfunc/mlp.py:114-145): output width = act_dim * pre_horizon.  `forward` is the eager
    definition for samplers; FHADP2 evaluates the same parameters through `gops_mlp_forward / _backward` (`hip_mlp()`)."""

    def __init__(self, **kwargs):
        super().__init__()
        self.act_dim = kwargs["act_dim"]
        self.pre_horizon = kwargs["pre_horizon"]
        sizes = [kwargs["obs_dim"]] + list(kwargs["hidden_sizes"]) + [self.act_dim * self.pre_horizon]
        self._hidden_activation = kwargs["hidden_activation"]
        self._output_activation = kwargs.get("output_activation", "linear")
        self.pi = mlp(sizes, get_activation_func(self._hidden_activation),
                      get_activation_func(self._output_activation))
        self.register_buffer("act_high_lim", torch.from_numpy(kwargs["act_high_lim"]).float())
        self.register_buffer("act_low_lim", torch.from_numpy(kwargs["act_low_lim"]).float())
        self.action_distribution_cls = kwargs["action_distribution_cls"]

    def pre_tanh(self, obs):
        """[B, pre_horizon, act_dim] head outputs before the tanh squash."""
        return self.pi(obs).reshape(obs.shape[0], self.pre_horizon, self.act_dim)

    def forward_all_policy(self, obs):
        y = self.pre_tanh(obs)
        return (self.act_high_lim - self.act_low_lim) / 2 * torch.tanh(y) + (self.act_high_lim + self.act_low_lim) / 2

    def forward(self, obs):
        return self.forward_all_policy(obs)[:, 0, :]


class StateValue(nn.Module, Action_Distribution, _HipMlpMixin):
    """State-value function: obs -> scalar."""

    _net_attr = "v"

    def __init__(self, **kwargs):
        super().__init__()
        self._hidden_activation = kwargs["hidden_activation"]
        self._output_activation = kwargs.get("output_activation", "linear")
        self.v = mlp([kwargs["obs_dim"]] + list(kwargs["hidden_sizes"]) + [1],
                     get_activation_func(self._hidden_activation),
                     get_activation_func(self._output_activation))
        self.action_distribution_cls = kwargs["action_distribution_cls"]

    def forward(self, obs):
        return torch.squeeze(self.v(obs), -1)


class ActionValue(nn.Module, Action_Distribution, _HipMlpMixin):
    """Action-value function: (obs, act) -> scalar over the concatenated input (reference gops/apprfunc/mlp.py:224-245,
    parameter names `q.0.weight` ...).  MPG evaluates it through `gops_mlp_forward / _backward(_x)` (`hip_mlp()`)."""

    _net_attr = "q"

    def __init__(self, **kwargs):
        super().__init__()
        self._hidden_activation = kwargs["hidden_activation"]
        self._output_activation = kwargs.get("output_activation", "linear")
        self.q = mlp([kwargs["obs_dim"] + kwargs["act_dim"]] + list(kwargs["hidden_sizes"]) + [1],
                     get_activation_func(self._hidden_activation),
                     get_activation_func(self._output_activation))
        self.action_distribution_cls = kwargs["action_distribution_cls"]

    def forward(self, obs, act):
        return torch.squeeze(self.q(torch.cat([obs, act], dim=-1)), -1)
