"""SPIL - separated proportional-integral Lagrangian - on the fused HIP rollout.

Reference: gops/algorithm/spil.py (ApproxContainer :32-66, SPIL :69-270).  Every update does BOTH
* policy evaluation (:189-212): V(o) regressed onto `sum_t gamma^t r_t + gamma^n V_target(o_n)` of a no-grad model
  rollout - the terminal value is NOT masked at done here (`tail_unmasked`) - while the rollout also yields, per
  constraint k, whether the trajectory stayed safe (`prod_t [c_tk <= 0]`); their batch mean is the safe probability;
* policy improvement (:214-251): ascend `w_r sum_t gamma^t r_t + sum_k w_c[k] prod_t Phi(c_tk)` through policy and
  model (no terminal value), with the weights from the PI(D) multiplier rule on `chance_thre - safe_prob` (:253-270,
  host arithmetic, kept verbatim).
The model must have constraint outputs: the kernels of pyth_veh3dofconti_surrcstr / _detour / _errcstr,
pyth_veh2dofconti_errcstr and pyth_mobilerobot return the Phi-products and safe flags next to v_pi
(`GopsRolloutOut.constraint_prods`) and take d(loss)/d(product) * product into the backward sweep
(`GopsRolloutIn.grad_constraint_prod`) - every script under example_train/spil/.
"""
__all__ = ["SPIL"]

import time
from typing import Any, Tuple

import numpy as np
import torch

from gops_amd import hip_backend as hb
from gops_amd.algorithm.base import _INFO_KEYS, AlgorithmBase, batch_to_device, cuda_device_of, grad_buffers
from gops_amd.algorithm.infadp import ApproxContainer  # noqa: F401  (same container: v, policy, targets, two Adams)
from gops_amd.create_pkg.create_env_model import create_env_model
from gops_amd.utils.tensorboard_setup import tb_tags


class SPIL(AlgorithmBase):
    """gamma, tau, pev_step, pim_step, forward_step as in the reference (spil.py:80-112)."""

    def __init__(self, index: int = 0, gamma: float = 0.99, tau: float = 0.005, pev_step: int = 1, pim_step: int = 1,
                 forward_step: int = 25, **kwargs: Any):
        super().__init__(index, **kwargs)
        self.networks = ApproxContainer(**kwargs)
        self.envmodel = create_env_model(**kwargs)
        self.gamma, self.tau = gamma, tau
        self.pev_step, self.pim_step, self.forward_step = pev_step, pim_step, forward_step
        self.reward_scale = 1.0
        self.n_constraint = kwargs["constraint_dim"]
        self.delta_i = np.array([0.0] * kwargs["constraint_dim"])
        self.Kp, self.Ki, self.Kd = 60, 0.02, 0
        self.tb_info = dict()
        self.safe_prob_pre = np.array([0.0] * kwargs["constraint_dim"])
        self.chance_thre = np.array([0.97] * kwargs["constraint_dim"])
        self._cache = {}

    @property
    def adjustable_parameters(self):
        return ("gamma", "tau", "pev_step", "pim_step", "forward_step", "reward_scale")

    def local_update(self, data: dict, iteration: int) -> dict:
        self._update(self._compute_gradient(data, iteration))
        return self.tb_info

    def get_remote_update_info(self, data: dict, iteration: int) -> Tuple[dict, dict]:
        update_list = self._compute_gradient(data, iteration)
        return self.tb_info, {name: [p.grad for p in self.networks.net_dict[name].parameters()] for name in update_list}

    def remote_update(self, update_info: dict):
        for net_name, grads in update_info.items():
            for p, grad in zip(self.networks.net_dict[net_name].parameters(), grads):
                p.grad = grad
        self._update(list(update_info.keys()))

    def _update(self, update_list: list):
        tau = self.tau
        for net_name in update_list:
            self.networks.optimizer_dict[net_name].step()
        with torch.no_grad():
            for net_name in update_list:
                online = list(self.networks.net_dict[net_name].parameters())
                target = list(self.networks.target_net_dict[net_name].parameters())
                torch._foreach_mul_(target, 1 - tau)
                torch._foreach_add_(target, online, alpha=tau)

    # ------------------------------------------------------------------------------------------
    def _rollout_for(self, batch: int, device, need_grad: bool) -> hb.Rollout:
        nets = self.networks
        key = (batch, self.forward_step, float(self.gamma), str(device), need_grad)
        pol = nets.policy.hip_mlp()
        vt = None if need_grad else nets.v_target.hip_mlp()   # PIM has no terminal value (spil.py:233-250)
        ro = self._cache.get(key)
        if ro is None:
            env = self.envmodel.hip_env(nets.policy.act_low_lim.cpu().numpy(), nets.policy.act_high_lim.cpu().numpy())
            if not hb.has_constraints(env):
                raise RuntimeError("SPIL needs a model with constraint outputs (pyth_veh3dofconti_surrcstr / _detour / _errcstr, "
                                   "pyth_veh2dofconti_errcstr, pyth_mobilerobot)")
            if env.n_constraint != self.n_constraint:
                raise RuntimeError(f"constraint_dim = {self.n_constraint}, but the model has {env.n_constraint} constraints")
            ro = hb.Rollout(env, pol, batch=batch, horizon=self.forward_step, gamma=self.gamma, finite_horizon=False,
                            need_grad=need_grad, value=vt, device=device, tail_unmasked=not need_grad)
            self._cache[key] = ro
        else:
            ro.set_policy(pol, vt)
        return ro

    def _value_for(self, batch: int, device) -> hb.ValueNet:
        key = ("v", batch, str(device))
        mlp = self.networks.v.hip_mlp()
        vn = self._cache.get(key)
        if vn is None:
            vn = self._cache[key] = hb.ValueNet(mlp, batch, device=device)
        else:
            vn.mlp = mlp
        return vn

    def _compute_gradient(self, data: dict, iteration: int) -> list:
        start_time = time.time()
        if self.reward_scale != 1.0:
            raise RuntimeError("SPIL.reward_scale != 1 is not supported by the HIP rollout (the terminal value is unscaled)")
        batch = self._attach_reference_points(data, batch_to_device(data, cuda_device_of(self.networks), ("obs", "done") + _INFO_KEYS))
        B, device, nc = batch["obs"].shape[0], batch["obs"].device, self.n_constraint
        # ---- policy evaluation -----------------------------------------------------------------
        res = self._rollout_for(B, device, need_grad=False).forward(batch)
        vn = self._value_for(B, device)
        v = vn.forward(batch["obs"])
        diff = v - res["v_pi"]
        gw, gb = grad_buffers(self.networks.v)
        vn.backward(batch["obs"], (2.0 / B) * diff, gw, gb)
        pev = torch.cat((torch.stack(((diff * diff).mean(), v.mean())), res["constraint_prods"][nc:].mean(1))).tolist()   # host sync
        self.tb_info[tb_tags["loss_critic"]], self.tb_info[tb_tags["critic_avg_value"]] = pev[0], pev[1]
        self.safe_prob = np.array(pev[2:], dtype=np.float32)
        # ---- policy improvement -----------------------------------------------------------------
        ro = self._rollout_for(B, device, need_grad=True)
        res = ro.forward(batch)
        w_r, w_c = self._spil_get_weight()
        w_c_t = torch.tensor(np.asarray(w_c, dtype=np.float32), device=device)
        c_mul = res["constraint_prods"][:nc]                                        # [nc, B]
        loss_pi = (w_r * res["v_pi"] + (c_mul * w_c_t[:, None]).sum(0)).mean()
        gw, gb = grad_buffers(self.networks.policy)
        grad_v = torch.full((B,), -float(w_r) / B, dtype=torch.float32, device=device)
        ro.backward(grad_v, gw, gb, grad_constraint_prod=(-(w_c_t[:, None] / B) * c_mul).contiguous())
        self.tb_info[tb_tags["loss_actor"]] = (-loss_pi).item()
        self.tb_info[tb_tags["alg_time"]] = (time.time() - start_time) * 1000
        return ["v", "policy"]

    def _spil_get_weight(self):   # spil.py:253-270, verbatim arithmetic
        delta_p = self.chance_thre - self.safe_prob
        # integral separation
        delta_p_sepa = np.where(np.abs(delta_p) > 0.1, delta_p * 0.7, delta_p)
        delta_p_sepa = np.where(np.abs(delta_p) > 0.2, delta_p * 0, delta_p_sepa)
        self.delta_i = np.clip(self.delta_i + delta_p_sepa, 0, 99999)
        delta_d = np.clip(self.safe_prob_pre - self.safe_prob, 0, 3333)
        lam = np.clip(self.Ki * self.delta_i + self.Kp * delta_p + self.Kd * delta_d, 0, 3333)
        self.safe_prob_pre = self.safe_prob
        self.lam = lam
        return 1 / (1 + lam.sum()), lam / (1 + lam.sum())
