"""Gradient exchange of the data-parallel trainer: one process per GPU, one collective per update.

The reference averages per-parameter gradient lists from N Ray actors in Python on the driver
(gops/trainer/off_sync_trainer.py:183-208, after `.cpu()`-ing them).  Here every rank keeps its
gradients on its MI355X, flattens them into ONE contiguous buffer and issues a single
all-reduce(sum) over RCCL/xGMI (backend "nccl"; "gloo" on CPU for the tests), then scales by 1/N.
The payload is 0.27-1.1 MB, i.e. latency-bound on the xGMI ring, so one collective per update
(not one per parameter) is what matters.
"""
from typing import Dict, List

import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _as_one_buffer(tensors: List[torch.Tensor]):
    """The 1-D tensor that `tensors` tile back to back (the algorithms allocate a network's gradients
    that way, `algorithm/base.py:grad_buffers`), or None if they are separate allocations."""
    if not tensors or not all(t.is_contiguous() for t in tensors):
        return None
    first = tensors[0]
    esz = first.element_size()
    ptr = first.data_ptr()
    for t in tensors:
        if t.dtype != first.dtype or t.device != first.device or t.data_ptr() != ptr:
            return None
        if t.untyped_storage().data_ptr() != first.untyped_storage().data_ptr():
            return None
        ptr += t.numel() * esz
    total = sum(t.numel() for t in tensors)
    return torch.as_strided(first, (total,), (1,), first.storage_offset())


class GradAllReducer:
    """In-place mean of `update_info` (dict: name -> list of gradient tensors) over all ranks."""

    def __init__(self, group=None, overlap: bool = True, single_rank_phases: bool = False):
        self.group = group
        self.overlap = overlap
        # measurement aid (bench.py --dp-path): with ONE rank, still let the algorithms take the N > 1 code path - two-phase
        # backward, start points, `_pending` - with nothing to exchange, so that the path's fixed cost can be timed on one GPU
        self.single_rank_phases = single_rank_phases
        self._works = []

    def overlap_enabled(self) -> bool:
        """More than one rank and not switched off: algorithms that can (`supports_overlapped_reduce`) start the all-reduce of
        the gradients that are ready first themselves (`start_`), behind the kernels that produced them."""
        return self.overlap and (world_size() > 1 or self.single_rank_phases)

    def start_(self, tensors: List[torch.Tensor]):
        """Asynchronous in-place SUM all-reduce of gradient tensors that tile ONE contiguous slice of a network's gradient
        buffer (`algorithm/base.py:grad_buffers`): the collective waits for what is queued on the current stream so far and runs
        on the process group's own stream - kernels queued afterwards overlap it.  `average_` waits for it.  Tensors that do not
        tile one buffer (gradients installed from outside) are reduced through a flattened copy - slower, never an error."""
        if world_size() == 1:   # (single_rank_phases: nothing to exchange)
            return
        flat = _as_one_buffer(tensors)
        if flat is None:
            flat = _flatten_dense_tensors(tensors)
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append((work, flat, list(tensors)))
            return
        self._works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, None))

    def abandon_(self):
        """Wait for and drop collectives that were started but will not be consumed (the caller's update raised between `start_`
        and `average_`): a later iteration must not wait on - or take the 1/N for - somebody else's handles."""
        works, self._works = self._works, []
        for w, _, _ in works:
            try:
                w.wait()
            except Exception:   # noqa: BLE001 - best effort while unwinding
                pass

    def average_(self, update_info: Dict[str, List[torch.Tensor]], defer_scale: bool = False) -> Dict[str, List[torch.Tensor]]:
        """`defer_scale`: leave the SUM in the buffers and hand the 1/N to the consumer as
        `update_info["_grad_scale"]` - the HIP Adam kernel multiplies it in as it reads the gradients
        (`hip_backend.HipAdam.grad_scale`), which saves the elementwise launch between the collective and
        the optimizer step.  Entries whose name starts with "_", and entries that are not lists of tensors, are left alone."""
        n = world_size()
        if n == 1:
            update_info.pop("_pending", None)
            return update_info
        if update_info.pop("_pending", False):   # the algorithm started the collectives itself (start_): only wait for them
            works, self._works = self._works, []   # (taken over first: whatever happens below, no stale handle survives this call)
            for w, flat, tensors in works:
                w.wait()                         # (NCCL / RCCL: the current stream waits, the host does not)
                if flat is not None:             # reduced through a flattened copy: scatter it back
                    torch._foreach_copy_(tensors, list(_unflatten_dense_tensors(flat, tensors)))
            if defer_scale:
                update_info["_grad_scale"] = 1.0 / n
            else:
                for name, grads in update_info.items():
                    if not name.startswith("_") and isinstance(grads, (list, tuple)):
                        for g in grads:
                            g.div_(n)
            return update_info
        if self._works:   # collectives of an update that never reached average_ (it raised): drain them before this one
            self.abandon_()
        # (MPG's update_info also carries its iteration counter, mpg.py:434: non-list entries are not gradients)
        tensors = [g for name in sorted(update_info)
                   if not name.startswith("_") and isinstance(update_info[name], (list, tuple)) for g in update_info[name]]
        flat = _as_one_buffer(tensors)
        in_place = flat is not None
        if not in_place:
            flat = _flatten_dense_tensors(tensors)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)   # SUM + scale: valid on every backend / RCCL build
        if defer_scale:
            update_info["_grad_scale"] = 1.0 / n
        else:
            flat.div_(n)
        if not in_place:
            torch._foreach_copy_(tensors, list(_unflatten_dense_tensors(flat, tensors)))
        return update_info

    def mean_scalar(self, value: float, device) -> float:
        if world_size() == 1:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return (t / world_size()).item()


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """Make every replica start from rank `src`'s weights (one flat broadcast)."""
    if world_size() == 1:
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    flat = _flatten_dense_tensors(tensors)
    dist.broadcast(flat, src=src)
    for dst, s in zip(tensors, _unflatten_dense_tensors(flat, tensors)):
        dst.copy_(s)
