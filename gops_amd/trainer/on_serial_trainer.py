"""Serial on-policy trainer: sample -> alg.local_update -> log / save / evaluate.

Constructor signature, kwargs keys, checkpoint naming (`apprfunc/apprfunc_{it}.pkl`,
`*_opt.pkl`) and TensorBoard tags follow the reference's gops/trainer/on_serial_trainer.py:30-152.
Differences that come from the MI355X design: the networks live on the GPU for the whole run
(the reference bounces them CPU<->GPU around every update with ModuleOnDevice, :81), the
evaluator is called in-process (no Ray), and logging is skipped when tensorboard is missing.
"""
import os
import time
from math import inf

import torch

from gops_amd.utils.tensorboard_setup import add_scalars, make_writer, tb_tags

__all__ = ["OnSerialTrainer"]


class _RunningMean:
    """Running mean of the sampler's tb dicts between two log points."""

    def __init__(self):
        self.data, self.n = {}, 0

    def add_average(self, d: dict):
        self.n += 1
        for k, v in d.items():
            self.data[k] = self.data.get(k, 0.0) + (v - self.data.get(k, 0.0)) / self.n

    def pop(self) -> dict:
        out, self.data, self.n = self.data, {}, 0
        return out


class OnSerialTrainer:
    def __init__(self, alg, sampler, evaluator, **kwargs):
        self.alg = alg
        self.sampler = sampler
        self.evaluator = evaluator
        self.networks = self.alg.networks
        if self.sampler is not None:
            self.sampler.networks = self.networks
        if kwargs.get("ini_network_dir") is not None:
            self.networks.load_state_dict(torch.load(kwargs["ini_network_dir"]))
        self.max_iteration = kwargs.get("max_iteration")
        self.log_save_interval = kwargs["log_save_interval"]
        self.apprfunc_save_interval = kwargs["apprfunc_save_interval"]
        self.eval_interval = kwargs["eval_interval"]
        self.best_tar = -inf
        self.save_folder = kwargs["save_folder"]
        self.iteration = 0
        self.last_eval_iteration = 0
        self.use_gpu = kwargs.get("use_gpu", True)
        self.writer = make_writer(self.save_folder) if self.save_folder else None
        add_scalars({tb_tags["alg_time"]: 0, tb_tags["sampler_time"]: 0}, self.writer, 0)
        self.sampler_tb_dict = _RunningMean()
        self.start_time = time.time()

    def _sample(self):
        samples, sampler_tb = self.sampler.sample_with_replay_format()
        self.sampler_tb_dict.add_average(sampler_tb)
        return samples

    def step(self):
        samples = self._sample()
        self.networks.train()
        alg_tb_dict = self.alg.local_update(samples, self.iteration)
        self.networks.eval()
        self._after_update(alg_tb_dict)

    def _after_update(self, alg_tb_dict):
        if self.iteration % self.log_save_interval == 0:
            print("Iter = ", self.iteration)
            add_scalars(alg_tb_dict, self.writer, step=self.iteration)
            add_scalars(self.sampler_tb_dict.pop(), self.writer, step=self.iteration)
        if self.iteration % self.apprfunc_save_interval == 0:
            self.save_apprfunc()
        if self.evaluator is not None and self.iteration - self.last_eval_iteration >= self.eval_interval:
            self._evaluate()

    def _evaluate(self):
        self.evaluator.load_state_dict({k: v.cpu() for k, v in self.networks.state_dict().items()})
        total_avg_return = self.evaluator.run_evaluation(self.iteration)
        self.last_eval_iteration = self.iteration
        if total_avg_return >= self.best_tar and self.iteration >= self.max_iteration / 5:
            self.best_tar = total_avg_return
            print("Best return = {}!".format(str(self.best_tar)))
            folder = os.path.join(self.save_folder, "apprfunc")
            os.makedirs(folder, exist_ok=True)
            for filename in os.listdir(folder):
                if filename.endswith("_opt.pkl"):
                    os.remove(os.path.join(folder, filename))
            torch.save(self.networks.state_dict(), os.path.join(folder, "apprfunc_{}_opt.pkl".format(self.iteration)))
        if self.writer is not None:
            self.writer.add_scalar(tb_tags["TAR of RL iteration"], total_avg_return, self.iteration)
            self.writer.add_scalar(tb_tags["TAR of total time"], total_avg_return, int(time.time() - self.start_time))
            self.writer.add_scalar(tb_tags["TAR of collected samples"], total_avg_return,
                                   self.sampler.get_total_sample_number())

    def train(self):
        while self.iteration < self.max_iteration:
            self.step()
            self.iteration += 1
        self.save_apprfunc()
        if self.writer is not None:
            self.writer.flush()

    def save_apprfunc(self):
        if not self.save_folder:
            return
        folder = os.path.join(self.save_folder, "apprfunc")
        os.makedirs(folder, exist_ok=True)
        torch.save(self.networks.state_dict(), os.path.join(folder, "apprfunc_{}.pkl".format(self.iteration)))
