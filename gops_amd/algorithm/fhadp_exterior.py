"""FHADPExterior - FHADP with an exterior (quadratic) penalty on the model's constraint outputs.

Same surface as the reference's gops/algorithm/fhadp_exterior.py:19-78:
    loss = -mean_b sum_t gamma^t r_t  +  penalty * mean_b sum_t gamma^t sum_k max(c_tk, 0)^2
with `penalty` multiplied by `penalty_increase` every `penalty_delay` updates up to `max_penalty`.  The rollout kernel
of the constrained veh3dofconti models (GOPS_ENV_VEH3DOF_SURR) returns the discounted constraint sums next to v_pi and
its backward sweep takes d(loss)/d(sums) per trajectory - here penalty / B on the quadratic sum."""
__all__ = ["FHADPExterior"]

from typing import Tuple

import torch

from gops_amd import hip_backend as hb
from gops_amd.algorithm.base import grad_buffers
from gops_amd.algorithm.fhadp import ApproxContainer, FHADP   # noqa: F401  (ApproxContainer: create_alg looks it up here)
from gops_amd.utils.lazy_scalar import scalar
from gops_amd.utils.tensorboard_setup import tb_tags


class ConstrainedFHADP(FHADP):
    """Shared by the Exterior / Interior / Lagrangian variants: forward rollout with constraint sums, per-trajectory
    weights of those sums, backward sweep, device scalars for the log."""

    LOG_KEYS: Tuple[str, ...] = ()

    def _coef_host(self) -> float:
        """Host value of the coefficient (penalty / multiplier) the NEXT gradient uses."""
        raise NotImplementedError

    def _constraint_terms(self, v_pi, cs, B, coef):
        """`coef`: the coefficient as a device tensor [1] -> (grad_constraint [3, B], stacked device scalars in
        LOG_KEYS order)."""
        raise NotImplementedError

    def _device_batch(self, data):
        # The coefficient travels with the batch as a device scalar: a captured update reads it from the graph's
        # static input (refreshed by the per-replay batch copy), so the schedule below keeps advancing under replay
        # and a schedule step does not invalidate the graph.
        batch = super()._device_batch(data)
        self._coef_used = float(self._coef_host())
        batch["_coef"] = torch.full((1,), self._coef_used, dtype=torch.float32, device=batch["obs"].device)
        return batch

    def _gradient_kernels(self, batch):
        B, device = batch["obs"].shape[0], batch["obs"].device
        ro = self._rollout_for(B, device)
        if not hb.has_constraints(ro.desc.env) or (ro.desc.env.n_surr > 0 and "surr_state" not in batch):
            raise RuntimeError(f"{type(self).__name__} needs a model with constraint outputs "
                               "(pyth_veh3dofconti_surrcstr / _detour / _errcstr) and its info in the batch")
        res = ro.forward(batch)
        gc, scalars = self._constraint_terms(res["v_pi"], res["constraint_sums"], B, batch["_coef"])
        gw, gb = grad_buffers(self.networks.policy)
        ro.backward(self._grad_v(B, device), gw, gb, grad_constraint=gc)
        return scalars

    def _fill_tb(self, out, lazy=False):
        for i, k in enumerate(self.LOG_KEYS):
            self.tb_info[k] = out[i] if lazy else scalar(out, i)
        self._fill_host_tb()

    def _fill_host_tb(self):
        pass


class FHADPExterior(ConstrainedFHADP):
    LOG_KEYS = (tb_tags["loss_actor"], tb_tags["loss_actor_reward"], tb_tags["loss_actor_constraint"])

    def __init__(self, *, pre_horizon: int, gamma: float = 1.0, penalty: float = 1.0, penalty_increase: float = 1.1,
                 penalty_delay: float = 100, max_penalty: float = 1e3, index: int = 0, **kwargs):
        super().__init__(pre_horizon=pre_horizon, gamma=gamma, index=index, **kwargs)
        self.penalty, self.penalty_increase = penalty, penalty_increase
        self.penalty_delay, self.max_penalty = penalty_delay, max_penalty
        self.update_step = 0

    @property
    def adjustable_parameters(self) -> Tuple[str]:
        return (*super().adjustable_parameters, "penalty", "penalty_increase", "penalty_delay")

    def _coef_host(self):
        return self.penalty

    def _constraint_terms(self, v_pi, cs, B, coef):
        loss_reward, loss_constraint = -v_pi.mean(), cs[0].mean()
        gc = torch.zeros(3, B, dtype=torch.float32, device=v_pi.device)
        gc[0] = coef / B
        return gc, torch.stack(((loss_reward + coef * loss_constraint).reshape(()), loss_reward, loss_constraint))

    def _after_gradient(self, out):   # fhadp_exterior.py:68-70
        self.update_step += 1
        if self.update_step % self.penalty_delay == 0:
            self.penalty = min(self.penalty * self.penalty_increase, self.max_penalty)

    def _fill_host_tb(self):   # the reference logs the coefficient AFTER the schedule step (fhadp_exterior.py:72-77)
        self.tb_info["Loss/Penalty coefficient-RL iter"] = self.penalty
