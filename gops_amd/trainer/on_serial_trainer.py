"""Serial on-policy trainer: sample -> alg.local_update -> log / save / evaluate
(constructor and kwargs of gops/trainer/on_serial_trainer.py:30-152; shared machinery in `_common.py`)."""
from gops_amd.trainer._common import TrainerBase, call_maybe_remote

__all__ = ["OnSerialTrainer"]


class OnSerialTrainer(TrainerBase):
    def _sample(self):
        self._refresh_sampler_networks()   # host samplers: fresh CPU copy of the weights (TrainerBase)
        samples, sampler_tb = call_maybe_remote(self.sampler, "sample_with_replay_format")
        self.sampler_tb_dict.add_average(sampler_tb)
        return samples

    def step(self):
        samples = self._sample()
        self.networks.train()
        alg_tb_dict = self.alg.local_update(samples, self.iteration)
        self.networks.eval()
        self._after_update(alg_tb_dict)
