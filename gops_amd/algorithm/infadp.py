"""INFADP - infinite-horizon approximate dynamic programming - on the fused HIP rollout.

Same class surface as the reference's gops/algorithm/infadp.py (ApproxContainer :31-64, INFADP
:67-213): alternating policy evaluation (PEV: regress V(o) onto the n-step model return plus the
bootstrapped target value, rollout without gradient) and policy improvement (PIM: ascend the
same quantity through policy, model and the target value's input), Adam + Polyak target update.
"""
__all__ = ["INFADP"]

import os
import time
from copy import deepcopy
from typing import Tuple

import torch
from gops_amd.utils.common_utils import make_adam

from gops_amd import hip_backend as hb
from gops_amd.algorithm.base import (_INFO_KEYS, AlgorithmBase, ApprBase, PrecisionGuard, batch_to_device, cuda_device_of,
                                     grad_buffers)
from gops_amd.utils.hip_graph import StepGraphCache
from gops_amd.utils.lazy_scalar import scalar
from gops_amd.create_pkg.create_apprfunc import create_apprfunc
from gops_amd.create_pkg.create_env_model import create_env_model
from gops_amd.utils.common_utils import get_apprfunc_dict
from gops_amd.utils.tensorboard_setup import tb_tags


class ApproxContainer(ApprBase):
    """Value + policy networks, frozen target copies, one Adam per online network.  The value
    network is constructed BEFORE the policy (RNG draw order of the reference, infadp.py:41-42)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        v_args = get_apprfunc_dict("value", **kwargs)
        policy_args = get_apprfunc_dict("policy", **kwargs)
        self.v = create_apprfunc(**v_args)
        self.policy = create_apprfunc(**policy_args)
        self.v_target = deepcopy(self.v)
        self.policy_target = deepcopy(self.policy)
        for p in list(self.v_target.parameters()) + list(self.policy_target.parameters()):
            p.requires_grad = False
        self.policy_optimizer = make_adam(self.policy.parameters(), lr=kwargs["policy_learning_rate"])
        self.v_optimizer = make_adam(self.v.parameters(), lr=kwargs["value_learning_rate"])
        self.net_dict = {"v": self.v, "policy": self.policy}
        self.target_net_dict = {"v": self.v_target, "policy": self.policy_target}
        self.optimizer_dict = {"v": self.v_optimizer, "policy": self.policy_optimizer}

    def create_action_distributions(self, logits):
        return self.policy.get_act_dist(logits)


class INFADP(AlgorithmBase):
    """forward_step: model rollout length; gamma; tau: Polyak factor; pev_step / pim_step:
    alternation period of evaluation and improvement."""

    def __init__(self, index=0, **kwargs):
        super().__init__(index, **kwargs)
        self.networks = ApproxContainer(**kwargs)
        self.envmodel = create_env_model(**kwargs)
        self.gamma = 0.99
        self.tau = 0.005
        self.pev_step = 1
        self.pim_step = 1
        self.forward_step = 10
        # arithmetic of the MLP contractions: "fp32" (exact, the 1e-4 parity path) or "fp16" (half-precision
        # MFMA, BASELINE.json configs[4])
        self.mlp_dtype = kwargs.get("mlp_dtype", "fp32")
        self.tb_info = dict()
        self._cache = {}
        self._graphs = {}
        self._polyak = {}   # net name -> hb.PolyakUpdater
        # measured rule for leaving the plane-split forward (algorithm/base.py PrecisionGuard), one per trained network
        self.precision_guard = {m: PrecisionGuard(kwargs.get("precision_check_interval"), kwargs.get("precision_threshold")) for m in ("v", "policy")}
        self._bufs = {}     # persistent device scratch: loss gradient, loss scalars (read them before the next update of the same mode)

    @property
    def adjustable_parameters(self):
        return ("gamma", "tau", "pev_step", "pim_step", "forward_step", "reward_scale")

    def local_update(self, data: dict, iteration: int) -> dict:
        # gradient + Adam + Polyak of the iteration's mode as one HIP graph when the update is launch-bound
        # (replay batches of the shipped examples are 64-256 samples); eager kernels otherwise
        start_time = time.time()
        batch = self._attach_reference_points(data, batch_to_device(data, cuda_device_of(self.networks), ("obs", "done") + _INFO_KEYS))
        mode = self._mode(iteration)
        opt = self.networks.optimizer_dict[mode]
        self._precision_check(mode, batch)

        # The plain algorithm: Adam step, Polyak step of the target (and, for PIM, the loss mean) ride on the backward's last launch
        # (ABI v12, gops_rollout_backward_update / gops_value_backward_update)
        fuse = (type(self)._gradient_kernels is INFADP._gradient_kernels and type(self)._update is INFADP._update
                and os.environ.get("GOPS_FUSED_UPDATE", "1") != "0")   # (host-side A/B knob)

        def update(b):
            if fuse:
                scalars, stepped = self._gradient_kernels(mode, b, fused_opt=opt)
                self._update([mode], optimizer_stepped=stepped)   # (stepped: Adam AND Polyak were part of the backward call)
                return scalars
            scalars = self._gradient_kernels(mode, b)
            self._update([mode])
            return scalars

        opt.grad_scale = 1.0   # (a data-parallel remote_update may have left 1/N behind)
        cache = self._graphs.setdefault(mode, StepGraphCache())
        scalars = cache.run(self._signature(mode, batch), batch, update, before_replay=opt.sync_hyper,
                            on_replay=opt.advance, work=batch["obs"].shape[0] * self.forward_step,
                            on_capture_fail=opt.resync_device_state)
        self._log(mode, scalars, start_time)
        return self.tb_info

    accepts_grad_scale = True   # remote_update honours update_info["_grad_scale"] (trainer/grad_sync.py)

    def get_remote_update_info(self, data: dict, iteration: int) -> Tuple[dict, dict]:
        # Data-parallel path: no host sync between the backward sweep and the gradient all-reduce - the loss
        # scalars stay on the device in tb_info and are read at log time.
        start_time = time.time()
        batch = self._attach_reference_points(data, batch_to_device(data, cuda_device_of(self.networks), ("obs", "done") + _INFO_KEYS))
        mode = self._mode(iteration)
        self._precision_check(mode, batch)
        scalars = self._gradient_kernels(mode, batch)
        if mode == "v":
            self.tb_info[tb_tags["loss_critic"]], self.tb_info[tb_tags["critic_avg_value"]] = scalars[0], scalars[1]
        else:
            self.tb_info[tb_tags["loss_actor"]] = scalars[0]
        self.tb_info[tb_tags["alg_time"]] = (time.time() - start_time) * 1000  # ms of host enqueue time
        update_info = {mode: [p.grad for p in self.networks.net_dict[mode].parameters()]}
        return self.tb_info, update_info

    def remote_update(self, update_info: dict):
        names = [k for k in update_info if not k.startswith("_")]
        for net_name in names:
            for p, grad in zip(self.networks.net_dict[net_name].parameters(), update_info[net_name]):
                p.grad = grad
            self.networks.optimizer_dict[net_name].grad_scale = float(update_info.get("_grad_scale", 1.0))
        self._update(names)
        for net_name in names:   # a later local_update on this object must not inherit the 1/N
            self.networks.optimizer_dict[net_name].grad_scale = 1.0

    def _update(self, update_list, optimizer_stepped=False):
        tau = self.tau
        if optimizer_stepped:   # Adam and Polyak step were part of the backward call (_gradient_kernels(fused_opt=))
            return
        for net_name in update_list:
            self.networks.optimizer_dict[net_name].step()
        with torch.no_grad():
            for net_name in update_list:   # Polyak averaging (reference :124-133), all tensors of the network in one launch
                online = list(self.networks.net_dict[net_name].parameters())
                target = list(self.networks.target_net_dict[net_name].parameters())
                if not online[0].is_cuda:
                    torch._foreach_mul_(target, 1 - tau)
                    torch._foreach_add_(target, online, alpha=tau)
                    continue
                pk = self._polyak.get(net_name)
                if pk is None or not pk.matches(target, online):
                    pk = self._polyak[net_name] = hb.PolyakUpdater(target, online)
                pk.step(tau)

    # ------------------------------------------------------------------------------------------
    def _rollout_for(self, batch: int, device, need_grad: bool) -> hb.Rollout:
        nets = self.networks
        flags = self._variant_flags("policy" if need_grad else "v")   # (PEV's gradient-free backup belongs to the value update)
        key = (batch, self.forward_step, float(self.gamma), str(device), need_grad, hb.dtype_id(self.mlp_dtype), flags)
        pol, vt = nets.policy.hip_mlp(), nets.v_target.hip_mlp()
        ro = self._cache.get(key)
        if ro is None:
            env = self.envmodel.hip_env(nets.policy.act_low_lim.cpu().numpy(), nets.policy.act_high_lim.cpu().numpy())
            ro = hb.Rollout(env, pol, batch=batch, horizon=self.forward_step, gamma=self.gamma,
                            finite_horizon=False, need_grad=need_grad, value=vt, device=device, dtype=self.mlp_dtype, variant_flags=flags)
            self._cache[key] = ro
        else:
            ro.set_policy(pol, vt)
        return ro

    def _value_for(self, batch: int, device) -> hb.ValueNet:
        flags = self._variant_flags("v")
        key = ("v", batch, str(device), hb.dtype_id(self.mlp_dtype), flags)
        mlp = self.networks.v.hip_mlp(self.mlp_dtype, variant_flags=flags)
        vn = self._cache.get(key)
        if vn is None:
            vn = self._cache[key] = hb.ValueNet(mlp, batch, device=device)
        else:
            vn.mlp = mlp
        return vn

    def _variant_flags(self, mode: str) -> int:
        forced = getattr(self, "_forced_flags", None)   # (set for the duration of a precision check)
        return self.precision_guard[mode].flags() if forced is None else forced

    def _precision_check(self, mode: str, batch):
        """PrecisionGuard (algorithm/base.py) of the network this iteration trains: every `interval` of ITS gradients the gradient of
        `batch` is formed with the launch's own kernels and with the exact-fp32 rollout kernels; beyond the threshold that network's
        launches stay on those."""
        guard = self.precision_guard[mode]
        if self.mlp_dtype != "fp32" or not PrecisionGuard.applies_to(self.networks.policy, self.networks.v,
                                                                      env_kind=getattr(self.envmodel.unwrapped, "hip_kind", None)):
            return
        if not guard.due():
            return

        def flat_gradient(flags):
            self._forced_flags = flags
            try:
                self._gradient_kernels(mode, batch)
            finally:
                self._forced_flags = None
            return self.networks.net_dict[mode]._flat_grad.clone()
        guard.check(flat_gradient)

    def _mode(self, iteration) -> str:
        return "v" if iteration % (self.pev_step + self.pim_step) < self.pev_step else "policy"

    def _gradient_kernels(self, mode: str, batch, fused_opt=None) -> torch.Tensor:
        """Enqueue one policy-evaluation ("v") or policy-improvement ("policy") gradient; returns the
        device scalars the log needs ([loss_v, mean V] / [loss_policy]) without synchronising."""
        B, device = batch["obs"].shape[0], batch["obs"].device
        if mode == "v":
            # PEV: loss_v = mean((V(o) - [sum_t gamma^t r_t + (~d) gamma^n V_target(o_n)])^2)
            backup = self._rollout_for(B, device, need_grad=False).forward(batch)["v_pi"]
            vn = self._value_for(B, device)
            v = vn.forward(batch["obs"])
            # loss_v, mean V and d(loss_v)/dV = (2 / B)(V - backup) in one launch (they were six torch passes)
            gdiff = self._scratch("gdiff", B, device)
            scalars = self._loss_stats("v", device).value_loss(v, backup, gdiff)
            gw, gb = grad_buffers(self.networks.v)
            if fused_opt is not None:   # -> (scalars, whether Adam + Polyak were part of the backward call)
                fa, pk = self._fused_parts("v", fused_opt)
                if fa is None:
                    vn.backward(batch["obs"], gdiff, gw, gb)
                    return scalars, False
                vn.backward(batch["obs"], gdiff, gw, gb, tail=hb.make_update_tail(fa, polyak=pk, tau=self.tau))
                fused_opt.end_fused()
                return scalars, True
            vn.backward(batch["obs"], gdiff, gw, gb)
            return scalars
        # PIM: loss = -mean(sum_t gamma^t r_t + (~d) gamma^n V_target(o_n)), grads into the policy
        ro = self._rollout_for(B, device, need_grad=True)
        v_pi = ro.forward(batch)["v_pi"]
        gw, gb = grad_buffers(self.networks.policy)
        if fused_opt is not None:   # -> (scalars, whether Adam + Polyak were part of the backward call)
            fa, pk = self._fused_parts("policy", fused_opt)
            stats = self._loss_stats("policy", device)
            if fa is None:
                ro.backward(self._grad_v(B, device), gw, gb)
                return stats.mean_loss(v_pi, -1.0)[:1], False
            ro.backward(self._grad_v(B, device), gw, gb, tail=hb.make_update_tail(fa, v_pi, -1.0, stats, polyak=pk, tau=self.tau))
            fused_opt.end_fused()
            return stats.buf[:1], True
        stats = self._loss_stats("policy", device)   # (the loss mean rides on the backward's reduce launch: it needs no gradient)
        ro.backward(self._grad_v(B, device), gw, gb, tail=hb.make_update_tail(None, v_pi, -1.0, stats))
        return stats.buf[:1]

    def _fused_parts(self, net_name, opt):
        """(what `HipAdam.begin_fused` returns, the network's one-table PolyakUpdater) for a backward call that carries the update's
        tail - or (None, None) when optimizer or target averaging do not fit one table each (the separate launches then)."""
        online = list(self.networks.net_dict[net_name].parameters())
        target = list(self.networks.target_net_dict[net_name].parameters())
        pk = self._polyak.get(net_name)
        if pk is None or not pk.matches(target, online):
            pk = self._polyak[net_name] = hb.PolyakUpdater(target, online)
        if len(pk.tables) != 1:
            return None, None
        fa = opt.begin_fused()
        return (fa, pk) if fa is not None else (None, None)

    def _scratch(self, name, n, device):
        t = self._bufs.get(name)
        if t is None or t.numel() != n or t.device != device:
            t = self._bufs[name] = torch.empty(n, dtype=torch.float32, device=device)
        return t

    def _loss_stats(self, mode, device) -> hb.LossStats:
        st = self._bufs.get(("stats", mode))
        if st is None or st.buf.device != device:
            st = self._bufs[("stats", mode)] = hb.LossStats(device)
        return st

    def _grad_v(self, B, device):
        gv = getattr(self, "_gv", None)
        if gv is None or gv.shape[0] != B or gv.device != device:
            gv = self._gv = torch.full((B,), -1.0 / B, dtype=torch.float32, device=device)
        return gv

    def _log(self, mode: str, scalars: torch.Tensor, start_time: float):
        # (LazyScalar: read back on first use; GOPS_EAGER_LOG=1: host sync here, as in the reference)
        if mode == "v":
            self.tb_info[tb_tags["loss_critic"]] = scalar(scalars, 0, on_value=self.precision_guard["v"].observe_loss)
            self.tb_info[tb_tags["critic_avg_value"]] = scalar(scalars, 1)
        else:
            self.tb_info[tb_tags["loss_actor"]] = scalar(scalars, 0, on_value=self.precision_guard["policy"].observe_loss)
        self.tb_info[tb_tags["alg_time"]] = (time.time() - start_time) * 1000  # ms

    def _compute_gradient(self, data, iteration):
        start_time = time.time()
        batch = self._attach_reference_points(data, batch_to_device(data, cuda_device_of(self.networks), ("obs", "done") + _INFO_KEYS))
        mode = self._mode(iteration)
        self._log(mode, self._gradient_kernels(mode, batch), start_time)
        return [mode]

    def _signature(self, mode, batch):
        nets = self.networks
        mods = (nets.v, nets.v_target, nets.policy) if mode == "v" else (nets.policy, nets.policy_target, nets.v_target)
        return (mode, tuple((k, tuple(v.shape)) for k, v in batch.items()), self.forward_step, float(self.gamma),
                float(self.tau), self._variant_flags(mode), self._variant_flags("v"),   # (a tripped guard forces a re-capture)
                tuple((p.data_ptr(), 0 if p.grad is None else p.grad.data_ptr()) for m in mods for p in m.parameters()),
                nets.optimizer_dict[mode].storage_signature(),   # Adam moments / device state, workspaces: raw pointers
                tuple(sorted(obj.workspace.data_ptr() for obj in self._cache.values())))
