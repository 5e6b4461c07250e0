"""HIP-graph replay of an algorithm's update step.

One ADP update is a fixed chain of ~10 kernels (prologue, forward rollout, loss mean, parameter
upload, backward sweep, weight-gradient GEMMs, reduce, Adam).  At small batches the chain is
launch-bound, and even at the target batch the inter-kernel gaps are ~5 % of the step, so after
`EAGER_CALLS` ordinary calls with an unchanged signature the chain is captured once into a HIP graph
(`torch.cuda.CUDAGraph` is hipGraph on ROCm) and replayed: the new batch is copied into the graph's
static input tensors (one foreach copy), one `hipGraphLaunch` runs the update.

Everything the captured kernels read besides the batch is either at a fixed device address
(parameters, gradients, Adam moments, workspace) or device-resident state (Adam's lr / step count),
so replays are exact repeats of the eager step.

Policy (`GOPS_HIP_GRAPH`): "auto" (default) captures only launch-bound steps - fewer than
`AUTO_MAX_WORK` env-model steps per update; measured on MI355X the graph is +40 % at B=64, H=10 and
-1 % at B=4096, H=30, where the host already runs ahead of the GPU and the extra batch copy is the
only difference.  "1" always captures, "0" never.
"""
import os
import warnings
from typing import Callable, Dict

import torch

EAGER_CALLS = 2   # eager calls with the same signature before capture (they also warm every kernel)


AUTO_MAX_WORK = 32768   # batch x horizon below which a step counts as launch-bound


def graphs_enabled(work: int = 0) -> bool:
    mode = os.environ.get("GOPS_HIP_GRAPH", "auto")
    if mode == "auto":
        return work < AUTO_MAX_WORK
    return mode != "0"


class GraphedStep:
    """`fn(batch) -> tensor` captured into a graph on static copies of `batch`."""

    def __init__(self, fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor], batch: Dict[str, torch.Tensor]):
        self.keys = list(batch.keys())
        self.static_in = {k: batch[k].clone() for k in self.keys}
        self._dst = [self.static_in[k] for k in self.keys]
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: a side thread may be talking to the runtime while this thread captures (the reference-point pipeline of
        # `strict_reference_points` evaluates the next batch and copies it over on its own stream, ref_traj_host.py) - under the
        # default "global" mode its synchronise / allocation calls would invalidate the capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out = fn(self.static_in)

    def __call__(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        torch._foreach_copy_(self._dst, [batch[k] for k in self.keys])
        self.graph.replay()
        # the graph's output buffer is rewritten by the next replay: callers keep results around unread (LazyScalar log
        # entries, utils/lazy_scalar.py), so every replay hands out its own copy (a few floats, no sync)
        return self.out.clone() if torch.is_tensor(self.out) else self.out


class StepGraphCache:
    """Keeps one captured graph per call-site; a change of `signature` (batch shape, horizon, gamma,
    parameter storage, ...) drops the graph and restarts the eager count."""

    def __init__(self):
        self.sig = None
        self.calls = 0
        self.graph = None
        self.failed = False

    def run(self, signature, batch, eager_fn, on_replay=None, before_replay=None, work=0, on_capture_fail=None):
        """Returns fn's output: eagerly for the first calls, by graph replay afterwards.
        `before_replay()` runs ahead of every replay (push changed hyper-parameters to the device),
        `on_replay()` after it (host-side bookkeeping the eager function would have done); `work` is
        the step's size (batch x horizon) for the "auto" policy; `on_capture_fail()` runs when a capture
        attempt raised (host-side bookkeeping done during the aborted capture must be re-synchronised)."""
        if not graphs_enabled(work) or self.failed:
            return eager_fn(batch)
        if signature != self.sig:
            self.sig, self.calls, self.graph = signature, 0, None
        if self.graph is None:
            if self.calls < EAGER_CALLS:
                self.calls += 1
                return eager_fn(batch)
            try:
                if before_replay is not None:
                    before_replay()
                self.graph = GraphedStep(eager_fn, batch)   # capture runs eager_fn's host side once
            except Exception as exc:   # capture unsupported in this setup: stay on eager HIP launches
                warnings.warn(f"HIP graph capture failed ({exc}); continuing with eager launches")
                self.failed, self.graph = True, None
                torch.cuda.synchronize()
                if on_capture_fail is not None:
                    on_capture_fail()
                return eager_fn(batch)
            return self.graph(batch)   # host bookkeeping for this step was done during capture
        if before_replay is not None:
            before_replay()
        out = self.graph(batch)
        if on_replay is not None:
            on_replay()
        return out
