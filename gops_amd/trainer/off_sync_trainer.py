"""Synchronous data-parallel OFF-policy trainer, re-partitioned for an 8 x MI355X node.

The reference's OffSyncTrainer (gops/trainer/off_sync_trainer.py:132-208) keeps N algorithm actors, N samplers and
N buffers behind Ray: every actor computes a gradient on a replay batch, the driver AVERAGES the N gradient lists in
Python (:183-208, after `.cpu()`-ing them) and applies the mean to the learner, then re-ships the state_dict to
every actor.  Here each of the N replicas is one process on its own GPU (`torchrun`; `torch.distributed` backend
"nccl" = RCCL over xGMI): rank r owns its sampler, its HBM-resident replay buffer and its copy of the networks;
per iteration it samples a replay batch locally, runs the fused rollout forward / backward, and the N gradients are
averaged with ONE flat SUM all-reduce (`grad_sync.GradAllReducer`; the 1/N rides inside the Adam kernel) before
every rank applies the identical optimizer step.  No weights travel after the initial broadcast, and no gradient or
batch crosses PCIe.  Buffer warm-up, `sample_interval`, checkpoints and logging follow `off_serial_trainer`
(rank 0 logs / saves / evaluates).
"""
import torch

from gops_amd.trainer.grad_sync import GradAllReducer, broadcast_parameters, rank, world_size
from gops_amd.trainer.off_serial_trainer import OffSerialTrainer

__all__ = ["OffSyncTrainer"]


class OffSyncTrainer(OffSerialTrainer):
    def __init__(self, alg, sampler, buffer, evaluator, **kwargs):
        # the reference passes lists of remote samplers / buffers: this rank takes its own
        if isinstance(sampler, (list, tuple)):
            sampler = sampler[rank() % len(sampler)]
        if isinstance(buffer, (list, tuple)):
            buffer = buffer[rank() % len(buffer)]
        if isinstance(alg, (list, tuple)):
            alg = alg[rank() % len(alg)]
        alg = alg.unwrap() if hasattr(alg, "unwrap") else alg   # create_alg's in-process actor handle
        super().__init__(alg, sampler, buffer, evaluator, **kwargs)
        self.reducer = GradAllReducer()
        if hasattr(self.alg, "set_lockstep_replicas"):   # every rank computes gradient k together: the precision guards may use a collective
            self.alg.set_lockstep_replicas(True)
        broadcast_parameters(self.networks, src=0)   # identical replicas (TrainerBase already moved them to the GPU)
        self._refresh_sampler_networks()
        self.is_chief = rank() == 0
        if not self.is_chief:   # only rank 0 logs, saves and evaluates
            self.writer = None
            self.evaluator = None
        self.num_replicas = world_size()

    def step(self):
        if self.iteration % self.sample_interval == 0:
            samples, sampler_tb = self._sampler_samples()
            self._store(samples)
            self.sampler_tb_dict.add_average(sampler_tb)
        replay_samples = self.buffer.sample_batch(self.replay_batch_size)
        self.networks.train()
        # no host sync between the backward sweep and the collective (loss scalars stay on the device until logged)
        alg_tb_dict, update_info = self._gradient(replay_samples)
        self.reducer.average_(update_info, defer_scale=getattr(self.alg, "accepts_grad_scale", False))
        self.alg.remote_update(update_info)
        self.networks.eval()
        self._after_update(alg_tb_dict)

    def _gradient(self, samples):
        """Local gradient; an algorithm that can (`supports_overlapped_reduce`) starts the all-reduce of the gradients that are
        ready first itself, overlapped with the rest of its backward (grad_sync.GradAllReducer.start_)."""
        if getattr(self.alg, "supports_overlapped_reduce", False):
            return self.alg.get_remote_update_info(samples, self.iteration, reducer=self.reducer)
        return self.alg.get_remote_update_info(samples, self.iteration)

    def save_apprfunc(self):
        if self.is_chief:
            super().save_apprfunc()
