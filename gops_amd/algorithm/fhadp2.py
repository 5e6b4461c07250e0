"""FHADP2 - finite-horizon ADP with an open-loop action sequence - on the HIP rollout.

Same class surface as the reference's gops/algorithm/fhadp2.py (ApproxContainer :26-46, FHADP2
:49-121): `FiniteHorizonFullPolicy` maps obs_0 to all H actions in ONE MLP evaluation, the env model is
rolled out with that sequence, `loss = -mean_b sum_t gamma^t r_t`.  The H-step env-model loop and its
autograd replay run in the fused kernels' open-loop mode (`gops_rollout_forward` with
`open_loop = 1`, `gops_rollout_backward_open_loop`); the single MLP evaluation and its backward run in the
library too (`gops_mlp_forward / _backward`: hidden stack on the rollout tiles, the act_dim * H wide output layer on its
own kernels) - no autograd graph, no rocBLAS.
"""
__all__ = ["FHADP2"]

import time
from typing import Tuple

import torch

from gops_amd import hip_backend as hb
from gops_amd.algorithm.base import (_INFO_KEYS, AlgorithmBase, ApprBase, batch_to_device, cuda_device_of,
                                     grad_buffers)
from gops_amd.create_pkg.create_apprfunc import create_apprfunc
from gops_amd.create_pkg.create_env_model import create_env_model
from gops_amd.utils.common_utils import get_apprfunc_dict, make_adam
from gops_amd.utils.lazy_scalar import scalar
from gops_amd.utils.tensorboard_setup import tb_tags


class ApproxContainer(ApprBase):
    """Approximate function container for FHADP2: one full-horizon policy + Adam (+ lr scheduler)."""

    def __init__(self, *, policy_learning_rate: float, **kwargs):
        super().__init__(**kwargs)
        self.policy = create_apprfunc(**get_apprfunc_dict("policy", **kwargs))
        self.policy_optimizer = make_adam(self.policy.parameters(), lr=policy_learning_rate)
        self.optimizer_dict = {"policy": self.policy_optimizer}
        self.init_scheduler(**kwargs)

    def create_action_distributions(self, logits):
        return self.policy.get_act_dist(logits)


class FHADP2(AlgorithmBase):
    """:param int pre_horizon: length of the emitted action sequence = model rollout horizon."""

    def __init__(self, *, pre_horizon: int, index: int = 0, **kwargs):
        super().__init__(index, **kwargs)
        self.networks = ApproxContainer(**kwargs, pre_horizon=pre_horizon)
        self.envmodel = create_env_model(**kwargs, pre_horizon=pre_horizon)
        self.forward_step = pre_horizon
        self.gamma = 1.0
        self.tb_info = dict()
        self._rollouts = {}

    @property
    def adjustable_parameters(self) -> Tuple[str]:
        return ("forward_step", "gamma")

    def _local_update(self, data, iteration: int):
        self._compute_gradient(data)
        self.networks.policy_optimizer.step()
        return self.tb_info

    def get_remote_update_info(self, data, iteration: int):
        self._compute_gradient(data)
        return self.tb_info, {"grad": [p._grad for p in self.networks.policy.parameters()]}

    def _remote_update(self, update_info):
        for p, grad in zip(self.networks.policy.parameters(), update_info["grad"]):
            p.grad = grad
        self.networks.policy_optimizer.step()

    def _rollout_for(self, batch: int, device) -> hb.Rollout:
        policy = self.networks.policy
        if self.forward_step != policy.pre_horizon:
            raise RuntimeError("FHADP2: forward_step must equal the policy's pre_horizon (its output width)")
        key = (batch, self.forward_step, float(self.gamma), str(device))
        ro = self._rollouts.get(key)
        if ro is None:
            env = self.envmodel.hip_env(policy.act_low_lim.cpu().numpy(), policy.act_high_lim.cpu().numpy())
            ro = hb.Rollout(env, None, batch=batch, horizon=self.forward_step, gamma=self.gamma,
                            finite_horizon=False, need_grad=True, device=device)
            self._rollouts[key] = ro
        return ro

    def _mlp_for(self, batch: int, device) -> hb.MlpNet:
        mlp = self.networks.policy.hip_mlp()
        key = ("mlp", batch, str(device))
        net = self._rollouts.get(key)
        if net is None:
            net = self._rollouts[key] = hb.MlpNet(mlp, batch, device=device)
        else:
            net.mlp = mlp
        return net

    def _compute_gradient(self, data):
        t0 = time.time()
        device = cuda_device_of(self.networks)
        batch = self._attach_reference_points(data, batch_to_device(data, device, ("obs", "done") + _INFO_KEYS))
        B = batch["obs"].shape[0]
        policy = self.networks.policy
        # ONE evaluation of the full-horizon policy emits all H actions (fhadp2.py:100-104): hidden stack on the rollout
        # tiles, wide output layer on its own kernels (gops_mlp_forward); no autograd graph, no rocBLAS
        net = self._mlp_for(B, device)
        pre = net.forward(batch["obs"]).view(B, policy.pre_horizon, policy.act_dim)
        ro = self._rollout_for(B, device)
        v_pi = ro.forward(batch, head_pre=pre)["v_pi"]
        loss_policy = -v_pi.mean()
        grad_v = torch.full((B,), -1.0 / B, dtype=torch.float32, device=device)
        g_pre = ro.backward_open_loop(grad_v)                    # d(loss)/d(head outputs) [B, H, A]
        gw, gb = grad_buffers(policy)
        net.backward(batch["obs"], g_pre.view(B, -1), gw, gb)    # through the MLP into the parameters' .grad
        self.tb_info[tb_tags["loss_actor"]] = scalar(loss_policy)
        self.tb_info[tb_tags["alg_time"]] = (time.time() - t0) * 1000  # ms
