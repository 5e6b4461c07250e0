"""Uniform replay buffer whose storage lives in HBM.

Interface of the reference's gops/trainer/buffer/replay_buffer.py:27-108 (`store`, `add_batch`,
`sample_batch`, `__len__`, `__get_RAM__`, keys `obs, obs2, act, rew, done, logp` plus `<k>` /
`next_<k>` for every `additional_info` key, everything float32 in the sampled batch, :105).  The
reference keeps numpy arrays on the host, indexes them per update and copies the batch to the GPU
inside the trainer (off_serial_trainer.py:91-93).  Here the ring buffers are device tensors: a sampled
batch is one `index_select` per key on the GPU and never crosses PCIe; `add_batch` stacks the new
transitions on the host once and issues one copy per key.  At 288 GB of HBM a 10^6-transition buffer
of veh3dof (obs 126 + info 136 floats, twice) is 2 GB.
"""
import numpy as np
import torch

from gops_amd.utils.common_utils import set_seed

__all__ = ["ReplayBuffer"]


def _shape(length, shape=None):
    if shape is None or shape == ():
        return (length,)
    return (length, shape) if np.isscalar(shape) else (length, *shape)


class ReplayBuffer:
    def __init__(self, index=0, **kwargs):
        set_seed(kwargs["trainer"], kwargs["seed"], index + 100)
        self.obsv_dim = kwargs["obsv_dim"]
        self.act_dim = kwargs["action_dim"]
        self.max_size = kwargs["buffer_max_size"]
        dev = kwargs.get("buffer_device")
        if dev is None:
            dev = "cuda" if (kwargs.get("use_gpu", True) and torch.cuda.is_available()) else "cpu"
        self.device = torch.device(dev)
        z = lambda shape=None: torch.zeros(_shape(self.max_size, shape), dtype=torch.float32, device=self.device)
        self.buf = {"obs": z(self.obsv_dim), "obs2": z(self.obsv_dim), "act": z(self.act_dim),
                    "rew": z(), "done": z(), "logp": z()}
        self.additional_info = kwargs.get("additional_info", {}) or {}
        for k, v in self.additional_info.items():
            if not isinstance(v, dict):
                raise NotImplementedError(f"additional_info['{k}']: only array-like entries ({{'shape', 'dtype'}}) "
                                          "are supported by the device buffer")
            self.buf[k] = z(v["shape"])
            self.buf["next_" + k] = z(v["shape"])
        self.ptr, self.size = 0, 0
        # sampling indices come from a device generator seeded like the reference seeds numpy
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(int(kwargs["seed"]) + index + 100)

    def __len__(self):
        return self.size

    def __get_RAM__(self):
        """MB occupied by the stored transitions (device memory here)."""
        per_row = sum(v[0].numel() * v.element_size() for v in self.buf.values())
        return per_row * self.size / 1e6

    def _rows(self, n):
        idx = (self.ptr + np.arange(n)) % self.max_size
        self.ptr = int((self.ptr + n) % self.max_size)
        self.size = int(min(self.size + n, self.max_size))
        return torch.as_tensor(idx, dtype=torch.long, device=self.device)

    def store(self, obs, act, rew, done, info, next_obs, next_info, logp):
        self.add_batch([(obs, act, rew, done, info, next_obs, next_info, logp)])

    def add_batch(self, samples: list):
        """samples: list of (obs, act, rew, done, info, next_obs, next_info, logp) as produced by the
        reference's samplers (sampler/base.py Experience tuples)."""
        n = len(samples)
        if n == 0:
            return
        if n > self.max_size:
            samples, n = samples[-self.max_size:], self.max_size
        rows = self._rows(n)

        def put(key, values):
            if self.buf[key][0].numel() == 0:   # a zero-size slot (pyth_mobilerobot declares "constraint" with shape (0,)):
                return                           # the reference's numpy buffer broadcasts the value into nothing
            host = torch.as_tensor(np.asarray(values, dtype=np.float32).reshape(_shape(n, tuple(self.buf[key].shape[1:]))))
            self.buf[key].index_copy_(0, rows, host.to(self.device, non_blocking=True))

        put("obs", [s[0] for s in samples])
        put("act", [s[1] for s in samples])
        put("rew", [s[2] for s in samples])
        put("done", [s[3] for s in samples])
        put("obs2", [s[5] for s in samples])
        put("logp", [s[7] for s in samples])
        for k in self.additional_info:
            put(k, [s[4][k] for s in samples])
            put("next_" + k, [s[6][k] for s in samples])

    def add_tensors(self, batch: dict):
        """Device-side insert of a dict of [n, ...] tensors with this buffer's keys (no host round trip):
        the entry point for samplers that already produce batched device data."""
        n = next(iter(batch.values())).shape[0]
        if n > self.max_size:   # (like add_batch: only the last max_size rows can survive - and no slot is written twice)
            batch, n = {k: v[-self.max_size:] for k, v in batch.items()}, self.max_size
        rows = self._rows(n)
        for k, dst in self.buf.items():
            if k in batch and dst[0].numel() > 0:
                dst.index_copy_(0, rows, batch[k].to(device=self.device, dtype=torch.float32).reshape(_shape(n, tuple(dst.shape[1:]))))

    def sample_batch(self, batch_size: int) -> dict:
        idx = torch.randint(0, self.size, (batch_size,), generator=self._gen, device=self.device)
        return {k: v.index_select(0, idx) for k, v in self.buf.items()}
