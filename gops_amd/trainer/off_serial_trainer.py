"""Serial off-policy trainer: sample -> buffer -> replay batch -> alg.local_update -> log / save / eval.

Constructor signature, kwargs keys, warm-up (`buffer_warm_size`), `sample_interval`, checkpoint naming
and TensorBoard tags follow the reference's gops/trainer/off_serial_trainer.py:30-188 - the trainer
every shipped FHADP / INFADP example script defaults to.  MI355X differences: the replay buffer is
device resident (`gops_amd/trainer/buffer/replay_buffer.py`), so the sampled batch is already in HBM
(the reference copies it over PCIe every step, :91-93); the networks stay on the GPU (the sampler gets
a CPU copy of the weights only when it samples, instead of the whole container bouncing with
ModuleOnDevice, :81); the evaluator runs in-process (no Ray).
"""
import time

from gops_amd.trainer._common import TrainerBase, call_maybe_remote

__all__ = ["OffSerialTrainer"]


class OffSerialTrainer(TrainerBase):
    def __init__(self, alg, sampler, buffer, evaluator, **kwargs):
        super().__init__(alg, sampler, evaluator, **kwargs)
        self.buffer = buffer
        # prioritized replay (off_serial_trainer.py:34-37, 96-100): the algorithm returns (tb_info, tree indices, new priorities)
        self.per_flag = kwargs.get("buffer_name") == "prioritized_replay_buffer"
        self.replay_batch_size = kwargs["replay_batch_size"]
        self.sample_interval = kwargs.get("sample_interval", 1)
        while self.buffer.size < kwargs["buffer_warm_size"]:   # pre sampling
            samples, _ = self._sampler_samples()
            self._store(samples)
        self.start_time = time.time()

    def _sampler_samples(self):
        self._refresh_sampler_networks()   # host samplers: fresh CPU copy of the weights (TrainerBase)
        return call_maybe_remote(self.sampler, "sample")

    def _store(self, samples):
        """Lists of Experience tuples (reference samplers) or a dict of batched device tensors
        (`DeviceEnvSampler`)."""
        if isinstance(samples, dict):
            self.buffer.add_tensors(samples)
        else:
            self.buffer.add_batch(samples)

    def step(self):
        if self.iteration % self.sample_interval == 0:
            samples, sampler_tb = self._sampler_samples()
            self._store(samples)
            self.sampler_tb_dict.add_average(sampler_tb)
        replay_samples = self.buffer.sample_batch(self.replay_batch_size)
        self.networks.train()
        if self.per_flag:   # (off_serial_trainer.py:96-100)
            out = self.alg.local_update(replay_samples, self.iteration)
            if not (isinstance(out, tuple) and len(out) == 3):
                raise RuntimeError(f"{type(self.alg).__name__}.local_update returns no (tb_info, idx, new_priority): this algorithm does "
                                   "not support buffer_name='prioritized_replay_buffer' (the reference's DSAC-family algorithms do)")
            alg_tb_dict, idx, new_priority = out
            self.buffer.update_batch(idx, new_priority)
        else:
            alg_tb_dict = self.alg.local_update(replay_samples, self.iteration)
        self.networks.eval()
        self._after_update(alg_tb_dict)

    def _evaluate(self):
        super()._evaluate()
        if self.writer is not None:
            from gops_amd.utils.tensorboard_setup import tb_tags
            self.writer.add_scalar(tb_tags["Buffer RAM of RL iteration"], self.buffer.__get_RAM__(), self.iteration)
