"""FHADPInterior - FHADP with a log-barrier on feasible trajectories and a quadratic penalty on infeasible ones.

Same surface as the reference's gops/algorithm/fhadp_interior.py:21-92: with feasible_b = [every c_tk < 0],
    loss = -mean v_r + (1 / penalty) * mean(feasible * sum_t gamma^t sum_k log(-min(c, 0) + 1e-8))
                     + penalty * mean(~feasible * sum_t gamma^t sum_k max(c, 0)^2).
The feasibility flags come out of the forward kernel with the sums, so the per-trajectory weights of the backward sweep
are formed on the device without a host round trip."""
__all__ = ["FHADPInterior"]

from typing import Tuple

import torch

from gops_amd.algorithm.base import grad_buffers
from gops_amd.algorithm.fhadp import ApproxContainer, FHADP   # noqa: F401  (ApproxContainer: create_alg looks it up here)
from gops_amd.utils.tensorboard_setup import tb_tags
from gops_amd.algorithm.fhadp_exterior import ConstrainedFHADP


class FHADPInterior(ConstrainedFHADP):
    LOG_KEYS = (tb_tags["loss_actor"], tb_tags["loss_actor_reward"], tb_tags["loss_actor_constraint"],
                "Loss/Feasible ratio-RL iter")

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
        feasible = cs[3]
        loss_reward = -v_pi.mean()
        loss_int = (cs[2] * feasible).mean()
        loss_ext = (cs[0] * (1.0 - feasible)).mean()
        gc = torch.zeros(3, B, dtype=torch.float32, device=v_pi.device)
        gc[2] = feasible * (1.0 / (coef * B))
        gc[0] = (1.0 - feasible) * (coef / B)
        loss = (loss_reward + (1.0 / coef) * loss_int + coef * loss_ext).reshape(())
        return gc, torch.stack((loss, loss_reward, loss_ext, feasible.mean()))

    def _after_gradient(self, out):   # fhadp_interior.py:80-82
        self.update_step += 1
        if self.update_step % self.penalty_delay == 0:
            self.penalty = min(self.penalty * self.penalty_increase, self.max_penalty)

    def _fill_host_tb(self):
        self.tb_info["Loss/Penalty coefficient-RL iter"] = self.penalty
