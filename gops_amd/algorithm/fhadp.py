"""FHADP - finite-horizon approximate dynamic programming - on the fused HIP rollout.

Same class surface as the reference's gops/algorithm/fhadp.py (ApproxContainer :32-55, FHADP
:58-125): `loss = -mean_b sum_t gamma^t r_t` over an H-step model rollout with the
FiniteHorizonPolicy, gradient into `networks.policy` parameters' `.grad`, Adam step.  The Python
loop `for step in range(H): a = policy(o, step+1); o, r, d, info = envmodel.forward(...)` and its
autograd replay are ONE forward and ONE backward kernel sweep (gops_rollout_forward/_backward).
"""
__all__ = ["FHADP"]

import os
import time
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
    """Approximate function container for FHADP: one policy network + Adam (+ lr scheduler)."""

    def __init__(self, *, policy_learning_rate: float, **kwargs):
        super().__init__(**kwargs)
        self.policy = create_apprfunc(**get_apprfunc_dict("policy", **kwargs))
        self.policy_optimizer = make_adam(self.policy.parameters(), lr=policy_learning_rate)
        self.optimizer_dict = {"policy": self.policy_optimizer}
        self.init_scheduler(**kwargs)

    def create_action_distributions(self, logits):
        return self.policy.get_act_dist(logits)


class FHADP(AlgorithmBase):
    """:param int pre_horizon: envmodel predict horizon.  :param float gamma: discount factor."""

    def __init__(self, *, pre_horizon: int, gamma: float = 1.0, index: int = 0, **kwargs):
        super().__init__(index, **kwargs)
        self.networks = ApproxContainer(**kwargs)
        self.envmodel = create_env_model(**kwargs, pre_horizon=pre_horizon)
        self.pre_horizon = pre_horizon
        self.gamma = gamma
        # arithmetic of the MLP contractions: "fp32" (exact, the 1e-4 parity path) or "fp16" (half-precision MFMA)
        self.mlp_dtype = kwargs.get("mlp_dtype", "fp32")
        self.tb_info = dict()
        self._rollouts = {}
        # measured rule for leaving the plane-split forward (algorithm/base.py PrecisionGuard); fp16 launches state their own tolerance
        self.precision_guard = PrecisionGuard(kwargs.get("precision_check_interval"), kwargs.get("precision_threshold"))
        self._update_graph, self._grad_graph, self._grad_graph_b = StepGraphCache(), StepGraphCache(), StepGraphCache()

    @property
    def adjustable_parameters(self) -> Tuple[str]:
        return ("pre_horizon", "gamma")

    def _local_update(self, data, iteration: int):
        # gradient + Adam as one captured HIP graph (eager for the first calls, see utils/hip_graph.py); the
        # Adam kernel is queued behind the backward pass BEFORE the host blocks on the loss scalar
        self._t0 = time.time()
        batch = self._device_batch(data)
        opt = self.networks.policy_optimizer
        self._precision_check(batch)

        # the constrained variants bring their own gradient kernels; the plain ones fold loss mean and Adam step into the backward's
        # last launch (ABI v12, gops_rollout_backward_update: two launches less per update)
        fuse = type(self)._gradient_kernels is FHADP._gradient_kernels and os.environ.get("GOPS_FUSED_UPDATE", "1") != "0"   # (host-side A/B knob)

        def update(b):
            if fuse:
                return self._gradient_kernels(b, fused_opt=opt)
            loss = self._gradient_kernels(b)
            opt.step()
            return loss

        opt.grad_scale = 1.0   # (a data-parallel remote_update may have left 1/N behind)
        loss = self._update_graph.run(self._signature(batch), batch, update,
                                      before_replay=opt.sync_hyper, on_replay=opt.advance,
                                      work=batch["obs"].shape[0] * self.pre_horizon,
                                      on_capture_fail=opt.resync_device_state)
        self._after_gradient(loss)   # host-side schedules: once per computed gradient, eager / captured / replayed alike
        self._log(loss)
        return self.tb_info

    accepts_grad_scale = True   # remote_update honours update_info["_grad_scale"] (trainer/grad_sync.py)

    supports_overlapped_reduce = True   # get_remote_update_info(..., reducer=...) starts the gradient all-reduce itself

    def get_remote_update_info(self, data, iteration: int, reducer=None):
        # Data-parallel path: NO host sync here - the gradient all-reduce is queued right behind the backward
        # sweep.  The loss stays a device scalar in tb_info (`add_scalars` reads it at log time).
        # With a `reducer` (trainer/grad_sync.GradAllReducer, more than one rank) the backward runs as two halves: everything
        # except the first hidden layer's weight gradient, whose all-reduce then travels on the collective's stream while the
        # first layer's GEMM + reduce still run on this one (GOPS_VF_BWD_PHASE_A / _B); `reducer.average_` later only waits.
        self._t0 = time.time()
        batch = self._device_batch(data)
        self._precision_check(batch)
        work = batch["obs"].shape[0] * self.pre_horizon
        grad_buffers(self.networks.policy)   # (allocates the flat gradient buffer on first use)
        grads = [p._grad for p in self.networks.policy.parameters()]
        info = {"grad": grads}
        plain = type(self)._gradient_kernels is FHADP._gradient_kernels   # (the constrained variants bring their own gradient kernels)
        if reducer is not None and reducer.overlap_enabled() and len(grads) >= 4 and plain:
            sig = self._signature(batch)
            loss = self._grad_graph.run(("a",) + sig, batch, lambda b: self._gradient_kernels(b, phase="a"), work=work)
            try:
                reducer.start_(grads[2:])   # output layer, hidden layers 1.. : final after phase A
                self._grad_graph_b.run(("b",) + sig, batch, lambda b: self._gradient_kernels(b, phase="b"), work=work)
                reducer.start_(grads[:2])   # first hidden layer
            except BaseException:
                reducer.abandon_()          # no handle of this update outlives it
                raise
            info["_pending"] = True
        else:
            loss = self._grad_graph.run(self._signature(batch), batch, self._gradient_kernels, work=work)
        self._after_gradient(loss)
        self._fill_tb(loss, lazy=True)                       # device scalars, read at log time
        self.tb_info[tb_tags["alg_time"]] = (time.time() - self._t0) * 1000   # ms of host enqueue time
        return self.tb_info, info

    def _remote_update(self, update_info):
        for p, grad in zip(self.networks.policy.parameters(), update_info["grad"]):
            p.grad = grad
        opt = self.networks.policy_optimizer
        opt.grad_scale = float(update_info.get("_grad_scale", 1.0))
        opt.step()
        opt.grad_scale = 1.0   # a later local_update on this object must not inherit the 1/N

    # ------------------------------------------------------------------------------------------
    def _rollout_for(self, batch: int, device) -> hb.Rollout:
        policy = self.networks.policy
        flags = self._variant_flags()
        key = (batch, self.pre_horizon, float(self.gamma), str(device), hb.dtype_id(self.mlp_dtype), flags)
        ro = self._rollouts.get(key)
        mlp = policy.hip_mlp()
        if ro is None:
            env = self.envmodel.hip_env(policy.act_low_lim.cpu().numpy(), policy.act_high_lim.cpu().numpy())
            ro = hb.Rollout(env, mlp, batch=batch, horizon=self.pre_horizon, gamma=self.gamma,
                            finite_horizon=True, need_grad=True, device=device, dtype=self.mlp_dtype, variant_flags=flags)
            # one live workspace per kernel variant of the current shape (shapes rarely change between updates)
            self._rollouts = {k: r for k, r in self._rollouts.items() if k[:5] == key[:5]}
            self._rollouts[key] = ro
        else:
            ro.set_policy(mlp)
        return ro

    def _variant_flags(self) -> int:
        forced = getattr(self, "_forced_flags", None)   # (set for the duration of a precision check)
        return self.precision_guard.flags() if forced is None else forced

    def _precision_check(self, batch):
        """PrecisionGuard (algorithm/base.py): every `interval` gradients the gradient of `batch` is formed with the launch's own
        kernels and with the exact-fp32 rollout kernels; beyond the threshold the algorithm stays on those."""
        if self.mlp_dtype != "fp32" or not PrecisionGuard.applies_to(self.networks.policy, env_kind=getattr(self.envmodel.unwrapped, "hip_kind", None)):
            return
        if not self.precision_guard.due():
            return

        def flat_gradient(flags):
            self._forced_flags = flags
            try:
                self._gradient_kernels(batch)
            finally:
                self._forced_flags = None
            return self.networks.policy._flat_grad.clone()
        self.precision_guard.check(flat_gradient)

    def _fill_tb(self, out, lazy=False):
        """`out` is what `_gradient_kernels` returned - here the pair [-mean(v_pi), mean(v_pi)] of `gops_mean_loss`; entry 0 is the
        loss.  lazy: leave a device tensor in tb_info (data-parallel path: a view, no launch); else a LazyScalar
        (utils/lazy_scalar.py: read back on first use; GOPS_EAGER_LOG=1: `.item()` right here, as the reference does)."""
        self.tb_info[tb_tags["loss_actor"]] = out[0] if lazy else scalar(out, 0, on_value=self.precision_guard.observe_loss)

    def _log(self, out):
        self._fill_tb(out)   # host sync, as in the reference
        self.tb_info[tb_tags["alg_time"]] = (time.time() - self._t0) * 1000  # ms

    def _device_batch(self, data):
        return self._attach_reference_points(data, batch_to_device(data, cuda_device_of(self.networks), ("obs", "done") + _INFO_KEYS))

    def _extra_signature(self):
        """Host-side values a subclass bakes into the captured kernels' inputs (penalty coefficients ...)."""
        return ()

    def _signature(self, batch):
        """What a captured graph is specialised on: shapes, rollout settings, parameter / gradient storage."""
        return (tuple((k, tuple(v.shape)) for k, v in batch.items()), self.pre_horizon, float(self.gamma), self._extra_signature(),
                self._variant_flags(),   # kernel variants are baked into a captured chain: a guard that trips must force a re-capture
                tuple((p.data_ptr(), 0 if p.grad is None else p.grad.data_ptr())
                      for p in self.networks.policy.parameters()),
                # everything else a captured kernel chain holds raw pointers to: Adam moments / device state
                # (replaced by optimizer.load_state_dict) and the rollout workspace (replaced when the shape changes)
                self.networks.policy_optimizer.storage_signature(),
                tuple(ro.workspace.data_ptr() for ro in self._rollouts.values()))

    def _gradient_kernels(self, batch, phase=None, fused_opt=None):
        """Enqueue forward rollout, backward sweep and the loss reduction; returns [-mean(v_pi), mean(v_pi)] (device tensor).  `phase`: None = all of it,
        "a" = everything but the first hidden layer's weight gradient, "b" = that gradient (hip_backend.Rollout.backward).
        `fused_opt` (the policy's HipAdam, single-process update): loss mean and optimizer step ride on the backward's last launch."""
        B, device = batch["obs"].shape[0], batch["obs"].device
        ro = self._rollout_for(B, device)
        if phase == "b":
            gw, gb = grad_buffers(self.networks.policy)
            ro.backward(self._grad_v(B, device), gw, gb, phase="b")
            return None
        v_pi = ro.forward(batch)["v_pi"]
        gw, gb = grad_buffers(self.networks.policy)
        if fused_opt is not None:
            fa = fused_opt.begin_fused()   # (p.grad of every parameter: grad_buffers' views into the flat buffer)
            if fa is not None:
                stats = self._loss_stats_slot(device)
                ro.backward(self._grad_v(B, device), gw, gb, tail=hb.make_update_tail(fa, v_pi, -1.0, stats))
                fused_opt.end_fused()
                return stats.buf[:2]
            ro.backward(self._grad_v(B, device), gw, gb)
            out = self._mean_of(v_pi)
            fused_opt.step()
            return out
        # loss = -mean(v_pi) (fhadp.py:121) is queued with every gradient - eager, captured or replayed -, so that an update always
        # contains the loss reduction the reference's `_compute_loss_policy` contains; it rides on the reduce launch of the backward
        # (of its phase A on the data-parallel path: the mean needs no gradient), one launch less than `gops_mean_loss` behind it
        stats = self._loss_stats_slot(device)
        ro.backward(self._grad_v(B, device), gw, gb, phase=phase, tail=hb.make_update_tail(None, v_pi, -1.0, stats))
        return stats.buf[:2]

    _LOSS_RING = 16

    def _loss_stats_slot(self, device):
        """The next of a small ring of persistent `LossStats` buffers (no allocation per update): a log entry (`LazyScalar`) read within
        `_LOSS_RING` updates of its own sees its own value."""
        ring = self.__dict__.setdefault("_loss_ring", [])
        if not ring or ring[0].buf.device != device:
            ring[:] = [hb.LossStats(device) for _ in range(self._LOSS_RING)]
            self._loss_slot = 0
        self._loss_slot = (self._loss_slot + 1) % len(ring)
        return ring[self._loss_slot]

    def _mean_of(self, v_pi):
        """[-mean(v_pi), mean(v_pi)] on the device (one `gops_mean_loss` launch into the ring's next buffer)."""
        return self._loss_stats_slot(v_pi.device).mean_loss(v_pi, -1.0)

    def _after_gradient(self, out):
        """Host-side bookkeeping that belongs to ONE computed gradient (penalty / multiplier schedules of the
        constrained variants).  Called by every update path after the gradient kernels were enqueued - as eager
        launches, during graph capture, or by graph replay - so a schedule advances exactly like the reference's,
        which steps it inside `_compute_loss_policy` (fhadp_exterior.py:68-70)."""

    def _compute_gradient(self, data, sync=True):
        self._t0 = time.time()
        loss_policy = self._gradient_kernels(self._device_batch(data))
        self._after_gradient(loss_policy)
        if sync:
            self._log(loss_policy)
        return loss_policy

    def _grad_v(self, B, device):
        """d(-mean v_pi)/d v_pi = -1/B for every trajectory (cached: it only depends on B)."""
        gv = getattr(self, "_gv", None)
        if gv is None or gv.shape[0] != B or gv.device != device:
            gv = self._gv = torch.full((B,), -1.0 / B, dtype=torch.float32, device=device)
        return gv
