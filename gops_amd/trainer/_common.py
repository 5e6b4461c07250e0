"""What every trainer of this package does around an update: running means of the sampler's log values,
periodic TensorBoard logging, checkpoints (`apprfunc/apprfunc_{it}.pkl`, best-so-far `*_opt.pkl`) and
in-process evaluation.  File names, intervals and tags follow the reference trainers
(gops/trainer/on_serial_trainer.py:30-152, off_serial_trainer.py:30-188); the networks live on the GPU
for the whole run and the evaluator is called directly (no Ray actor)."""
import copy
import os
import time
from math import inf

import torch

from gops_amd.utils.tensorboard_setup import add_scalars, make_writer, tb_tags


class RunningMean:
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


def call_maybe_remote(obj, name, *args):
    """`obj.name(*args)`, or `ray.get(obj.name.remote(*args))` when `obj` is a Ray actor handle: GOPS's
    `create_evaluator` returns `ray.remote(Evaluator).remote(...)` (gops/create_pkg/create_evaluator.py),
    whose methods cannot be called directly."""
    method = getattr(obj, name)
    if hasattr(method, "remote"):
        import ray
        return ray.get(method.remote(*args))
    return method(*args)


class TrainerBase:
    def __init__(self, alg, sampler, evaluator, **kwargs):
        self.alg, self.sampler, self.evaluator = alg, sampler, evaluator
        self.networks = self.alg.networks
        if kwargs.get("ini_network_dir") is not None:
            self.networks.load_state_dict(torch.load(kwargs["ini_network_dir"]))
        # The learner's networks live on the MI355X from construction on (the algorithms have no CPU path): every
        # pointer the HIP path caches stays valid, and device samplers see device weights from the first call.
        if torch.cuda.is_available() and not next(self.networks.parameters()).is_cuda:
            self.networks.to(torch.device("cuda", torch.cuda.current_device()))
        self._host_networks = None
        # a Ray actor handle (the reference's create_sampler for the sync / async trainers, create_sampler.py:71-81): its
        # methods go through `.remote()`, and it receives the weights as a state_dict instead of sharing the container
        self._sampler_remote = self.sampler is not None and hasattr(getattr(self.sampler, "sample", None), "remote")
        if self.sampler is not None:
            if not self._sampler_remote:
                self.sampler.networks = self.networks
            self._refresh_sampler_networks()
        self.max_iteration = kwargs.get("max_iteration")
        self.log_save_interval = kwargs["log_save_interval"]
        self.apprfunc_save_interval = kwargs["apprfunc_save_interval"]
        self.eval_interval = kwargs["eval_interval"]
        self.save_folder = kwargs["save_folder"]
        self.use_gpu = kwargs.get("use_gpu", True)
        self.best_tar = -inf
        self.iteration = 0
        self.last_eval_iteration = 0
        self.writer = make_writer(self.save_folder) if self.save_folder else None
        add_scalars({tb_tags["alg_time"]: 0, tb_tags["sampler_time"]: 0}, self.writer, 0)
        self.sampler_tb_dict = RunningMean()
        self.start_time = time.time()

    def _refresh_sampler_networks(self):
        """Samplers that step numpy envs with the policy on the CPU (the reference's, INTEGRATION.md option A') get
        their own host copy of the container, refreshed from the learner's weights before each sampling call (a
        0.4 MB device-to-host copy); the learner's parameters never move.  Device samplers (`on_device`) share the
        learner's container."""
        if self.sampler is None:
            return
        if self._sampler_remote:   # (sampler/base.py:80-81: `self.networks.load_state_dict(state_dict)` on the actor)
            call_maybe_remote(self.sampler, "load_state_dict", {k: v.cpu() for k, v in self.networks.state_dict().items()})
            return
        if getattr(self.sampler, "on_device", False):
            return
        if next(self.networks.parameters()).is_cuda:
            if self._host_networks is None:
                self._host_networks = copy.deepcopy(self.networks).to("cpu")
                self.sampler.networks = self._host_networks
            self._host_networks.load_state_dict(self.networks.state_dict())

    # ---- one iteration = subclass `step()`, then this -------------------------------------------
    def _warn_skipped_steps(self):
        """Log points only (one host read per optimizer): the optimizer kernels skip gradient elements that are not finite
        (`GopsAdamState.skipped_nonfinite`) - say so instead of training on silently."""
        seen = self.__dict__.setdefault("_skipped_seen", {})
        for name, opt in getattr(self.networks, "optimizer_dict", {}).items():
            n = opt.skipped_nonfinite() if hasattr(opt, "skipped_nonfinite") else 0
            if n > seen.get(name, 0):
                import warnings
                warnings.warn(f"gops_amd: {n - seen.get(name, 0)} non-finite gradient element(s) of '{name}' took no optimizer step since the "
                              f"last log point (iteration {self.iteration}); weights and moments of those elements are unchanged")
                seen[name] = n

    def _after_update(self, alg_tb_dict):
        if self.iteration % self.log_save_interval == 0:
            print("Iter = ", self.iteration)
            self._warn_skipped_steps()
            add_scalars(alg_tb_dict, self.writer, step=self.iteration)
            add_scalars(self.sampler_tb_dict.pop(), self.writer, step=self.iteration)
        if self.iteration % self.apprfunc_save_interval == 0:
            self.save_apprfunc()
        if self.evaluator is not None and self.iteration - self.last_eval_iteration >= self.eval_interval:
            self._evaluate()

    def _apprfunc_dir(self):
        folder = os.path.join(self.save_folder, "apprfunc")
        os.makedirs(folder, exist_ok=True)
        return folder

    def _evaluate(self):
        call_maybe_remote(self.evaluator, "load_state_dict", {k: v.cpu() for k, v in self.networks.state_dict().items()})
        total_avg_return = call_maybe_remote(self.evaluator, "run_evaluation", self.iteration)
        self.last_eval_iteration = self.iteration
        if total_avg_return >= self.best_tar and self.iteration >= self.max_iteration / 5:
            self.best_tar = total_avg_return
            print("Best return = {}!".format(str(self.best_tar)))
            folder = self._apprfunc_dir()
            for filename in os.listdir(folder):
                if filename.endswith("_opt.pkl"):
                    os.remove(os.path.join(folder, filename))
            torch.save(self.networks.state_dict(), os.path.join(folder, "apprfunc_{}_opt.pkl".format(self.iteration)))
        if self.writer is not None:
            self.writer.add_scalar(tb_tags["TAR of RL iteration"], total_avg_return, self.iteration)
            self.writer.add_scalar(tb_tags["TAR of total time"], total_avg_return, int(time.time() - self.start_time))
            self.writer.add_scalar(tb_tags["TAR of collected samples"], total_avg_return,
                                   call_maybe_remote(self.sampler, "get_total_sample_number"))

    def train(self):
        while self.iteration < self.max_iteration:
            self.step()
            self.iteration += 1
        self.save_apprfunc()
        if self.writer is not None:
            self.writer.flush()

    def save_apprfunc(self):
        if self.save_folder:
            torch.save(self.networks.state_dict(),
                       os.path.join(self._apprfunc_dir(), "apprfunc_{}.pkl".format(self.iteration)))
