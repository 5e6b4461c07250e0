"""Asynchronous data-parallel OFF-policy trainer, re-partitioned for an 8 x MI355X node.

The reference's OffAsyncTrainer (gops/trainer/off_async_trainer.py:35-200) keeps a "center network" on the driver and N
algorithm / sampler / buffer actors behind Ray: whenever an actor's gradient task completes, the driver applies that
(possibly stale) gradient to the center network (`self.networks.remote_update(update_info)`, :153-170), ships the
fresh state_dict back to that actor and hands it the next replay batch - no actor ever waits for another one.

Here every actor is one process on its own GPU (`torchrun`; `torch.distributed`, backend "nccl" = RCCL point-to-point over
xGMI; "gloo" in the CPU tests).  Rank 0 is the center network: it owns the authoritative weights and the optimizer
state, applies gradients in the order they ARRIVE - a worker bumps its arrival counter in the process group's key-value
store before it sends, the center reads all counters with one `multi_get` per iteration (~0.1 ms) and receives only from
workers whose gradient is on the way, so it never blocks on a slow rank and never uses a collective - and answers each
with the fresh flat weights (0.3-1.1 MB over one xGMI link).  (Polling `irecv(...).is_completed()` would be the obvious
form; gloo only completes a receive inside `wait()`, so the arrival notice goes through the store on every backend.)  It also
works its own GPU: between arrivals it computes and applies gradients of its own replay batches (those are never stale).
Ranks 1..N-1 loop: sample -> own HBM replay buffer -> replay batch -> gradient on their current weights -> send it ->
receive fresh weights.  A gradient is therefore at most one round trip old, as in the reference.  `max_iteration` counts
applied gradients (the reference's `self.iteration`); rank 0 logs, saves and evaluates.

Wire format (one fixed-size fp32 message per direction, so both sides can post their receives up front):
    worker -> center : [key style | presence flag per gradient slot | presence flag + value per scalar slot |
                        one region per trainable network of the container (flat, parameters() order; zeros when absent)]
    center -> worker : [stop flag, every parameter and buffer of the container (flat)]
A gradient slot is a trainable child network of the container; `update_info` may name it "<net>" (INFADP / SPIL: any
subset per iteration, SPIL ships "v" AND "policy"), "<net>_grad" (MPG: q1 / q2 / [q1_model / q2_model] / policy) or
"grad" (FHADP: the policy); scalar entries (MPG's "iteration") travel in the header.  The layout is derived from the
container alone, so it is identical on every rank before the first message.
"""
import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors

from gops_amd.trainer.grad_sync import broadcast_parameters, rank, world_size
from gops_amd.trainer.off_serial_trainer import OffSerialTrainer

__all__ = ["OffAsyncTrainer"]


class OffAsyncTrainer(OffSerialTrainer):
    def __init__(self, alg, sampler, buffer, evaluator, **kwargs):
        if isinstance(sampler, (list, tuple)):
            sampler = sampler[rank() % len(sampler)]
        if isinstance(buffer, (list, tuple)):
            buffer = buffer[rank() % len(buffer)]
        if isinstance(alg, (list, tuple)):
            alg = alg[rank() % len(alg)]
        alg = alg.unwrap() if hasattr(alg, "unwrap") else alg   # create_alg's in-process actor handle
        super().__init__(alg, sampler, buffer, evaluator, **kwargs)
        broadcast_parameters(self.networks, src=0)
        self._refresh_sampler_networks()
        self.is_center = rank() == 0
        if not self.is_center:
            self.writer = None
            self.evaluator = None
        self.n = world_size()
        self._state = [p.data for p in self.networks.parameters()] + [b.data for b in self.networks.buffers()]
        device = self._state[0].device
        # gradient slots: every child network with trainable parameters, in a fixed (sorted) order shared by all ranks.  A slot
        # covers ALL parameters of its network - the algorithms build their gradient lists from `mod.parameters()` and
        # `remote_update` zips against the same list, frozen entries included (a None gradient travels as zeros)
        self._slots = []      # (net name, parameters, offset of its region in the message body)
        off = 0
        for name, mod in sorted(self.networks.named_children()):
            params = list(mod.parameters())
            if any(p.requires_grad for p in params):
                self._slots.append((name, params, off))
                off += sum(p.numel() for p in params)
        self._slot_of = {name: i for i, (name, _, _) in enumerate(self._slots)}
        self._scalar_names = ("iteration",)
        self._head = 1 + len(self._slots) + 2 * len(self._scalar_names)   # [key style | slot flags | (flag, value) per scalar]
        self._grad_msg = torch.zeros(self._head + off, dtype=torch.float32, device=device)
        self._weight_msg = torch.zeros(1 + sum(t.numel() for t in self._state), dtype=torch.float32, device=device)
        self._stop = False
        self.local_iteration = 0          # this rank's own gradient count (drives INFADP's PEV / PIM alternation)
        self.applied_from = [0] * self.n  # center: gradients applied per source rank
        self._kv = dist.distributed_c10d._get_default_store() if self.n > 1 else None
        self._sent = 0                    # worker: gradients announced so far
        self._seen = [0] * self.n         # center: gradients received per worker
        if self.n > 1:
            self._kv.add(f"gops_async_arrivals_{rank()}", 0)   # every counter exists before anyone reads it
            dist.barrier()
        if self.is_center and self.n > 1:
            self._inbox = torch.zeros_like(self._grad_msg)

    # ---- message packing ----------------------------------------------------------------------
    def _slot_index(self, key: str) -> int:
        for cand in (key, key[:-5] if key.endswith("_grad") else None, "policy" if key == "grad" else None):
            if cand is not None and cand in self._slot_of:
                return self._slot_of[cand]
        raise ValueError(f"off_async_trainer: update_info entry '{key}' names no trainable network of "
                         f"{type(self.networks).__name__} (slots: {[n for n, _, _ in self._slots]})")

    def _pack_grad(self, update_info):
        """Every gradient list and every known scalar of `update_info` (entries starting with "_" are local hints)."""
        msg = self._grad_msg
        msg.zero_()
        keys = [k for k, v in update_info.items() if not k.startswith("_") and isinstance(v, (list, tuple))]
        # how this algorithm class names its gradient lists: "grad" (FHADP), "<net>_grad" (MPG) or "<net>"
        msg[0] = 2.0 if keys == ["grad"] else (1.0 if keys and all(k.endswith("_grad") for k in keys) else 0.0)
        for key, val in update_info.items():
            if key.startswith("_"):
                continue
            if isinstance(val, (list, tuple)):
                i = self._slot_index(key)
                _, params, off = self._slots[i]
                if len(val) != len(params):
                    raise ValueError(f"off_async_trainer: '{key}' carries {len(val)} tensors, the network has {len(params)}")
                flat = _flatten_dense_tensors([(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(val, params)])
                msg[1 + i] = 1.0
                msg[self._head + off:self._head + off + flat.numel()] = flat
            elif key in self._scalar_names:
                j = 1 + len(self._slots) + 2 * self._scalar_names.index(key)
                msg[j], msg[j + 1] = 1.0, float(val)
            else:
                raise ValueError(f"off_async_trainer: cannot ship update_info['{key}'] ({type(val).__name__})")
        return msg

    def _unpack_grad(self, msg):
        head = msg[:self._head].tolist()   # one host read for style, flags and scalars
        style = int(round(head[0]))
        info = {}
        for i, (name, params, off) in enumerate(self._slots):
            if head[1 + i] != 0.0:
                n = sum(p.numel() for p in params)
                grads = _unflatten_dense_tensors(msg[self._head + off:self._head + off + n], [p.data for p in params])
                info[{0: name, 1: name + "_grad", 2: "grad"}[style]] = [g.clone() for g in grads]
        for k, sname in enumerate(self._scalar_names):
            j = 1 + len(self._slots) + 2 * k
            if head[j] != 0.0:
                info[sname] = int(round(head[j + 1]))   # (fp32 header: exact up to 2^24 iterations)
        return info

    def _pack_weights(self, stop: bool):
        self._weight_msg[0] = 1.0 if stop else 0.0
        self._weight_msg[1:] = _flatten_dense_tensors([t.reshape(-1) for t in self._state])
        return self._weight_msg

    def _load_weights(self, msg):
        for dst, src in zip(self._state, _unflatten_dense_tensors(msg[1:], self._state)):
            dst.copy_(src)
        return bool(msg[0].item() != 0.0)

    # ---- one local gradient ---------------------------------------------------------------------
    def _local_gradient(self):
        if self.local_iteration % self.sample_interval == 0:
            samples, sampler_tb = self._sampler_samples()
            self._store(samples)
            self.sampler_tb_dict.add_average(sampler_tb)
        replay_samples = self.buffer.sample_batch(self.replay_batch_size)
        self.networks.train()
        alg_tb_dict, update_info = self.alg.get_remote_update_info(replay_samples, self.local_iteration)
        self.networks.eval()
        self.local_iteration += 1
        return alg_tb_dict, update_info

    def _apply(self, update_info, alg_tb_dict, src):
        self.alg.remote_update(update_info)
        self.applied_from[src] += 1
        self.iteration += 1
        self._after_update(alg_tb_dict)

    # ---- center / worker loops ----------------------------------------------------------------------
    def step(self):
        if not self.is_center:
            _, update_info = self._local_gradient()
            msg = self._pack_grad(update_info)
            self._sent += 1
            self._kv.add(f"gops_async_arrivals_{rank()}", 1)   # arrival notice, then the payload
            dist.send(msg, dst=0)
            dist.recv(self._weight_msg, src=0)
            self._stop = self._load_weights(self._weight_msg)
            return
        # center: first every gradient that is on its way (one store round trip tells which), then one of its own
        for w in self._arrived():
            if self.iteration < self.max_iteration:
                self._serve(w, stop=False)
        if self.iteration < self.max_iteration:
            alg_tb_dict, update_info = self._local_gradient()
            self._apply(update_info, alg_tb_dict, 0)

    def _arrived(self):
        if self.n == 1:
            return []
        counts = self._kv.multi_get([f"gops_async_arrivals_{w}" for w in range(1, self.n)])
        return [w for w, c in zip(range(1, self.n), counts) if int(c) > self._seen[w]]

    def _serve(self, w: int, stop: bool):
        """Receive worker w's announced gradient; apply it (unless stopping) and answer with the current weights."""
        dist.recv(self._inbox, src=w)
        self._seen[w] += 1
        if not stop:
            self._apply(self._unpack_grad(self._inbox), {}, w)
        dist.send(self._pack_weights(stop=stop), dst=w)

    def train(self):
        if self.is_center:
            while self.iteration < self.max_iteration:
                self.step()
            for w in range(1, self.n):   # every worker's next gradient is answered with "stop" + the final weights
                self._serve(w, stop=True)
            self.save_apprfunc()
            if self.writer is not None:
                self.writer.flush()
        else:
            while not self._stop:
                self.step()

    def save_apprfunc(self):
        if self.is_center:
            super().save_apprfunc()
