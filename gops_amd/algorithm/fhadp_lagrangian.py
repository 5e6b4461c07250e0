"""FHADPLagrangian - FHADP with a learned Lagrange multiplier on the summed constraint violation.

Same surface as the reference's gops/algorithm/fhadp_lagrangian.py:22-85:
    loss = -mean v_r + softplus(multiplier_param) * mean_b sum_t gamma^t sum_k max(c_tk, 0),
and every `multiplier_delay` updates one Adam ascent step on `multiplier_param` with the current mean violation
(a scalar: it stays a host-side torch parameter, as in the reference)."""
__all__ = ["FHADPLagrangian"]

import math
from typing import Tuple

from torch import nn
from torch.optim import Adam

import torch

from gops_amd.algorithm.base import grad_buffers
from gops_amd.algorithm.fhadp import ApproxContainer, FHADP   # noqa: F401  (ApproxContainer: create_alg looks it up here)
from gops_amd.utils.tensorboard_setup import tb_tags
from gops_amd.algorithm.fhadp_exterior import ConstrainedFHADP


class FHADPLagrangian(ConstrainedFHADP):
    LOG_KEYS = (tb_tags["loss_actor"], tb_tags["loss_actor_reward"], tb_tags["loss_actor_constraint"])

    def __init__(self, *, pre_horizon: int, gamma: float = 1.0, multiplier: float = 1.0, multiplier_lr: float = 1e-3,
                 multiplier_delay: int = 10, index: int = 0, **kwargs):
        super().__init__(pre_horizon=pre_horizon, gamma=gamma, index=index, **kwargs)
        # inverse of the softplus function
        self.multiplier_param = nn.Parameter(torch.tensor(math.log(math.exp(multiplier) - 1), dtype=torch.float32))
        self.multiplier_optim = Adam([self.multiplier_param], lr=multiplier_lr)
        self.multiplier_lr, self.multiplier_delay = multiplier_lr, multiplier_delay
        self.update_step = 0

    @property
    def multiplier(self) -> float:
        return torch.nn.functional.softplus(self.multiplier_param).item()

    @multiplier.setter
    def multiplier(self, value: float):
        with torch.no_grad():
            self.multiplier_param.fill_(math.log(math.exp(value) - 1))

    @property
    def adjustable_parameters(self) -> Tuple[str]:
        return (*super().adjustable_parameters, "multiplier", "multiplier_lr", "multiplier_delay")

    def _coef_host(self):
        return self.multiplier

    def _constraint_terms(self, v_pi, cs, B, coef):
        loss_reward, loss_constraint = -v_pi.mean(), cs[1].mean()
        gc = torch.zeros(3, B, dtype=torch.float32, device=v_pi.device)
        gc[1] = coef / B
        return gc, torch.stack(((loss_reward + coef * loss_constraint).reshape(()), loss_reward, loss_constraint))

    def _after_gradient(self, out):   # fhadp_lagrangian.py:72-77 (the host reads the mean violation only every `multiplier_delay` updates)
        self.update_step += 1
        if self.update_step % self.multiplier_delay == 0:
            # out[2] = mean violation of THIS gradient: the eager result, or the captured graph's static output
            multiplier_loss = -self.multiplier_param * out[2].item()
            self.multiplier_optim.zero_grad()
            multiplier_loss.backward()
            self.multiplier_optim.step()

    def _fill_host_tb(self):   # the reference logs the multiplier the loss was formed with (fhadp_lagrangian.py:70, 83)
        self.tb_info["Loss/Lagrange multiplier-RL iter"] = self._coef_used
