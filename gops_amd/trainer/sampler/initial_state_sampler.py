"""A minimal on-device batch source for the ADP trainers.

FHADP/INFADP read only the initial condition of each trajectory from a batch (`obs`, `done` and
the env info keys; fhadp.py:114-115, infadp.py:160-168).  This sampler draws those initial
conditions from the data environments' reset distributions (gops_amd.utils.synthetic) and keeps
them resident on the GPU.  It offers the two calls the trainers use
(`sample_with_replay_format`, `get_total_sample_number`).  The reference's closed-loop samplers
(gops/trainer/sampler/*, CPU numpy env stepping) are out of scope of this hot path.
"""
import time

import torch

from gops_amd.utils.synthetic import make_batch
from gops_amd.utils.tensorboard_setup import tb_tags


class InitialStateSampler:
    def __init__(self, cfg: dict, seed: int = 0, device=None):
        self.cfg, self.seed, self.count = cfg, seed, 0
        self.device = device
        self.networks = None
        self.total = 0

    def sample_with_replay_format(self):
        t0 = time.time()
        batch = make_batch(self.cfg, self.seed + self.count)
        self.count += 1
        self.total += batch["obs"].shape[0]
        if self.device is not None:
            batch = {k: v.to(self.device, non_blocking=True) for k, v in batch.items()}
        return batch, {tb_tags["sampler_time"]: (time.time() - t0) * 1000}

    def get_total_sample_number(self):
        return self.total

    def load_state_dict(self, state_dict):
        pass
