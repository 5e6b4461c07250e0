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
