"""Base classes of the algorithms: the approximate-function container and the update API the
trainers call (interface of the reference's gops/algorithm/base.py:24-121)."""
from abc import ABCMeta, abstractmethod
from typing import Dict, Tuple

import torch

from gops_amd.utils.common_utils import set_seed


class ApprBase(torch.nn.Module):
    """nn.Module container of the networks; wires optional `<net>_scheduler` kwargs of the form
    {"name": <torch.optim.lr_scheduler class name>, "params": {...}} to the optimizers."""

    def __init__(self, **kwargs):
        super().__init__()
        if kwargs.get("cnn_shared", False):
            raise NotImplementedError("cnn_shared feature networks are outside the MI355X ADP path")

    def init_scheduler(self, **kwargs):
        assert hasattr(self, "optimizer_dict")
        self.scheduler_dict = {}
        for key in [k for k in kwargs if k.endswith("_scheduler")]:
            cls = getattr(torch.optim.lr_scheduler, kwargs[key]["name"])
            self.scheduler_dict[key] = cls(self.optimizer_dict[key.replace("_scheduler", "")],
                                           **kwargs[key]["params"])


class AlgorithmBase(metaclass=ABCMeta):
    """index: offset of the random seed for sub-processes (sync/async trainers)."""

    def __init__(self, index, **kwargs):
        self.networks = None
        set_seed(kwargs["trainer"], kwargs["seed"], index + 300)
        # opt-in: the appended reference points of the veh3dofconti / veh2dofconti rollouts come from the host's torch CPU ops -
        # the reference's own values on this host - instead of the kernels' correctly rounded evaluation
        # (env/env_ocp/resources/ref_traj_host.py); no effect for models without reference trajectories
        self.strict_reference_points = bool(kwargs.get("strict_reference_points", False))
        self._ref_pipeline = None

    # ---- strict reference points (ref_traj_host.py) -----------------------------------------------------------------------------
    def _rollout_horizon(self) -> int:
        """Steps of this algorithm's model rollout (= reference points a rollout appends per trajectory)."""
        return int(getattr(self, "pre_horizon", None) or getattr(self, "forward_step"))

    def _reference_pipeline(self):
        """The `ReferencePointPipeline` of this algorithm's env model, or None (mode off / a model without reference trajectories)."""
        if not self.strict_reference_points:
            return None
        if self._ref_pipeline is None:
            from gops_amd import hip_backend as hb
            from gops_amd.env.env_ocp.resources.ref_traj_host import HostRefTraj, ReferencePointPipeline
            m = self.envmodel.unwrapped
            if getattr(m, "hip_kind", None) not in (hb.ENV_VEH, hb.ENV_VEH_SURR, hb.ENV_VEH2DOF):
                self._ref_pipeline = False
            else:
                self._ref_pipeline = ReferencePointPipeline(HostRefTraj(getattr(m, "ref_c", None), dt=m.dt), m.pre_horizon)
        return self._ref_pipeline or None

    def precision_guards(self):
        g = getattr(self, "precision_guard", None)
        return [] if g is None else (list(g.values()) if isinstance(g, dict) else [g])

    def set_lockstep_replicas(self, lockstep: bool = True, group=None) -> None:
        """Called by trainers whose ranks compute every gradient together (on_sync / off_sync): the precision guards then take
        their decision with a collective over `group`.  Never called by the asynchronous trainer."""
        for g in self.precision_guards():
            g.lockstep, g.group = bool(lockstep), group

    def prefetch_reference_points(self, data: dict) -> None:
        """Start evaluating the appended reference points of `data` (a batch a later update / gradient call will receive - the
        same dict, or one holding the same `ref_time` tensor) on the side thread.  No-op unless `strict_reference_points`."""
        pipe = self._reference_pipeline()
        if pipe is not None and data.get("ref_appended") is None:
            pipe.request(data, self._rollout_horizon(), cuda_device_of(self.networks))

    def _attach_reference_points(self, data: dict, batch: dict) -> dict:
        """`batch` (device tensors of `data`) with `ref_appended` [B, H, 4] in strict mode: the caller's own, the prefetched, or
        evaluated on the spot."""
        device = batch["obs"].device
        pts = data.get("ref_appended")   # a caller's own table is honoured in either mode
        if pts is None:
            pipe = self._reference_pipeline()
            if pipe is None:
                return batch
            pts = pipe.collect(data, self._rollout_horizon(), device)
        batch["ref_appended"] = pts.to(device=device, dtype=torch.float32).contiguous()
        return batch

    @property
    @abstractmethod
    def adjustable_parameters(self) -> tuple:
        ...

    def set_parameters(self, param_dict):
        for key in param_dict:
            if hasattr(self, key) and key in self.adjustable_parameters:
                setattr(self, key, param_dict[key])
            else:
                raise RuntimeError("param '" + key + "'is not adjustable in algorithm!")

    def get_parameters(self):
        return {p: getattr(self, p) for p in self.adjustable_parameters}

    def state_dict(self):
        return self.networks.state_dict()

    def load_state_dict(self, state_dict):
        self.networks.load_state_dict(state_dict)

    def _step_schedulers(self):
        for scheduler in getattr(self.networks, "scheduler_dict", {}).values():
            scheduler.step()

    def local_update(self, data: dict, iteration: int) -> dict:
        tb_info = self._local_update(data, iteration)
        self._step_schedulers()
        return tb_info

    def remote_update(self, update_info: dict):
        self._remote_update(update_info)
        self._step_schedulers()

    def _local_update(self, data: dict, iteration: int) -> dict:
        pass

    def get_remote_update_info(self, data: dict, iteration: int) -> Tuple[dict, dict]:
        raise NotImplementedError

    def _remote_update(self, update_info: dict):
        raise NotImplementedError

    def to(self, device):
        self.networks.to(device)

    def train(self):
        self.networks.train()

    def eval(self):
        self.networks.eval()


class PrecisionGuard:
    """When must a launch leave the plane-split kernels for exact fp32 products?  A measured rule.

    The default 256-wide kernels carry every operand of a hidden-layer contraction as two half planes (22 significant bits,
    csrc/common.h GOPS_SPLIT_F16X2), i.e. they evaluate the network with weights moved by up to 2^-22 relative - the same
    perturbed network for every sample, so this part of the error does not average out over the batch.  On every fixture trained
    by the reference (tests/test_trained256_gpu.py) that is at the level of the exact-fp32 kernels (2e-6 .. 3e-5 against the
    reference; the round-3 planes, 2^-20, put an ill-conditioned closed loop - the pyth_lq policy with a saturated tanh head, whose
    REFERENCE gradient moves 6.6e-5 under 1-ulp weight moves - at 3.5e-4), and activations beyond the half range of the forward
    planes (|a| >= 1.05e6) make the launch return NaN.  Neither conditioning nor range can be read off a description - so the guard
    MEASURES: every `interval` gradients (and at the first one: a loaded checkpoint may already be there) the gradient of the
    current batch is formed twice, with the launch's own kernels and with `exact_rollout_flags()` added - forward rollout and sweep on
    exact fp32 MFMAs; the weight-gradient GEMM keeps its two-half-plane products, whose per-sample operand errors average out over
    the B x H samples (exact rollout kernels + that GEMM measure the same as an all-exact launch on every trained fixture) - and if
    their relative L2 distance exceeds `threshold` the algorithm stays on the exact-fp32 rollout kernels from then on (sticky;
    0.59x the default's rate at the BASELINE target).  Cost: two extra gradients per `interval` updates (0.5 % at the default 500),
    one host sync per check.  `GOPS_PRECISION_CHECK_INTERVAL=0` (or `precision_check_interval=0`) switches the guard off."""

    def __init__(self, interval=None, threshold=None):
        import os
        self.interval = int(os.environ.get("GOPS_PRECISION_CHECK_INTERVAL", 500)) if interval is None else int(interval)
        self.threshold = float(os.environ.get("GOPS_PRECISION_THRESHOLD", 5e-5)) if threshold is None else float(threshold)
        self.count = 0            # gradients seen
        self.exact = False        # sticky: the rollout kernels run on exact fp32 products from here on
        self.last_distance = None
        self.checks = 0
        self.forced = False       # a logged loss was not finite: check the next gradient, whatever the schedule says
        # Replicas that update in LOCKSTEP (on_sync / off_sync trainers: every rank computes gradient k at the same time) take the
        # decision together - MAX all-reduce of the measured distance over `group`.  Anything else (a single process, the
        # asynchronous trainer whose ranks keep their own gradient counts and talk point to point) decides locally: a collective
        # there would wait for ranks that never enter it.  Set by the trainer (`AlgorithmBase.set_lockstep_replicas`), never
        # inferred from torch.distributed being initialised.
        self.lockstep = False
        self.group = None

    def observe_loss(self, value: float) -> None:
        """Hook of the algorithms' lazily read loss scalars: a non-finite loss (the plane-split kernels answer a half-range overflow
        with NaN; the optimizer kernels skip non-finite gradient elements, so the weights are still intact) makes the next
        gradient a checked one."""
        if not (value == value and abs(value) != float("inf")) and not self.exact and self.interval > 0:
            self.forced = True

    @staticmethod
    def applies_to(*modules, env_kind=None) -> bool:
        """Is there anything to guard?  Plane-split kernels exist for networks whose hidden layers are all 256 wide
        (csrc/rollout_fwd.hip: split_eligible / ss_shape_ok) - narrower nets run exact fp32 products anyway -, and two gradient
        evaluations of the same batch are only comparable when the model is deterministic (pyth_mobilerobot draws its obstacle
        noise per rollout)."""
        from gops_amd import hip_backend as hb
        if env_kind == hb.ENV_MOBILEROBOT:
            return False
        # (a net without a hidden layer has nothing to split: all([]) must not count as "every hidden layer is 256 wide")
        return any(len(m.linear_layers()) > 1 and all(l.out_features == 256 for l in m.linear_layers()[:-1]) for m in modules)

    @staticmethod
    def exact_rollout_flags():
        from gops_amd import hip_backend as hb
        return hb.VF_NO_STATIONARY_SPLIT | hb.VF_NO_STREAMED_SPLIT_FWD | hb.VF_NO_STREAMED_SPLIT_VALUE

    def flags(self) -> int:
        """Variant flags of the owner's launches right now (on top of hip_backend.DEFAULT_VARIANT_FLAGS)."""
        from gops_amd import hip_backend as hb
        return hb.DEFAULT_VARIANT_FLAGS | (self.exact_rollout_flags() if self.exact else 0)

    def due(self) -> bool:
        """Call once per computed gradient; True when this one is to be checked."""
        self.count += 1
        if self.exact or self.interval <= 0 or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            return False
        if self.forced and not self.lockstep:   # (lockstep replicas keep to the common schedule: a forced check on one rank only would hang the collective)
            self.forced = False
            return True
        return self.count == 1 or self.count % self.interval == 0

    def check(self, flat_gradient) -> float:
        """`flat_gradient(flags) -> 1-D device tensor`: the owner's gradient of the batch at hand under the given variant flags
        (a fresh tensor).  Returns the measured distance and switches to the exact rollout kernels when it exceeds the threshold."""
        from gops_amd import hip_backend as hb
        base = hb.DEFAULT_VARIANT_FLAGS
        g_split = flat_gradient(base)
        g_exact = flat_gradient(base | self.exact_rollout_flags())
        d = (g_split.double() - g_exact.double()).norm() / g_exact.double().norm().clamp_min(1e-300)
        if self.lockstep:   # replicas in lockstep take the decision together (the largest distance any rank saw; NaN from any rank wins)
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
                d = torch.where(torch.isfinite(d), d, torch.full_like(d, float("inf")))
                dist.all_reduce(d, op=dist.ReduceOp.MAX, group=self.group)
        self.last_distance = float(d.item())
        self.checks += 1
        if not (self.last_distance <= self.threshold):   # (NaN counts as exceeded)
            self.exact = True
            import warnings
            warnings.warn(f"gops_amd: plane-split and exact-fp32 rollout kernels differ by {self.last_distance:.2e} (relative L2 of the "
                          f"gradient, threshold {self.threshold:.0e}) - this network stays on the exact-fp32 rollout kernels from here on")
        return self.last_distance


# ---- helpers shared by the HIP-backed algorithms ---------------------------------------------
_INFO_KEYS = ("state", "ref_points", "path_num", "u_num", "ref_time", "surr_state")


def batch_to_device(data: Dict[str, torch.Tensor], device, keys) -> Dict[str, torch.Tensor]:
    """fp32, contiguous, on `device` - the layout the C ABI expects (bool `done` and uint8 ids
    from on-policy samplers are widened here, like the replay buffer does: replay_buffer.py:105)."""
    out = {}
    for k in keys:
        v = data.get(k)
        if v is not None:
            if v.dtype is not torch.float32 or v.device != device or not v.is_contiguous():
                v = v.to(device=device, dtype=torch.float32, non_blocking=True).contiguous()
            out[k] = v
    return out


def cuda_device_of(networks) -> torch.device:
    """The algorithms compute on the MI355X only: move the container there on first use."""
    p = next(networks.parameters())
    if not p.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("gops_amd algorithms need an MI355X (no CPU path); torch.cuda is unavailable")
        networks.to(torch.device("cuda", torch.cuda.current_device()))
        p = next(networks.parameters())
    return p.device


def grad_buffers(module):
    """Per-Linear-layer (weight grads, bias grads).  The `.grad` tensors of one network are views into
    ONE flat buffer (allocated here on first use, in `parameters()` order), so that the data-parallel
    trainer can all-reduce a network's gradient in place without flattening / copying back
    (`trainer/grad_sync.py`).  Gradients installed from outside (`remote_update`) are kept as they are."""
    params = list(module.parameters())
    flat = getattr(module, "_flat_grad", None)
    if any(p.grad is None or not p.grad.is_contiguous() for p in params):
        total = sum(p.numel() for p in params)
        flat = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
        off = 0
        for p in params:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        module._flat_grad = flat
    gw, gb = [], []
    for layer in module.linear_layers():
        gw.append(layer.weight.grad)
        gb.append(layer.bias.grad)
    return gw, gb
