"""Synchronous data-parallel on-policy trainer, re-partitioned for an 8 x MI355X node.

The reference's OnSyncTrainer (gops/trainer/on_sync_trainer.py:33-194) fans out N Ray sampler
actors, CONCATENATES their batches and runs one learner on the N x B batch.  Here the same
global update is computed by N replicas, one process per GPU (`torchrun`, `torch.distributed`
backend "nccl" = RCCL over xGMI): each rank draws its own B samples, runs the fused rollout
forward/backward locally, and the gradients are averaged with ONE flat all-reduce
(`grad_sync.GradAllReducer`).  The mean over ranks of per-rank batch-mean gradients equals the
gradient of the concatenated batch (equal B per rank), so the update is the reference's update;
every rank then applies the identical optimizer step - no weight broadcast per iteration
(the reference re-ships the state_dict to every sampler each step, :86-88).
"""
import torch

from gops_amd.trainer.grad_sync import GradAllReducer, broadcast_parameters, rank, world_size
from gops_amd.trainer.on_serial_trainer import OnSerialTrainer

__all__ = ["OnSyncTrainer"]


class OnSyncTrainer(OnSerialTrainer):
    def __init__(self, alg, sampler, evaluator, **kwargs):
        # `sampler` is this rank's sampler (the reference passes the list of remote samplers)
        if isinstance(sampler, (list, tuple)):
            sampler = sampler[rank() % len(sampler)]
        super().__init__(alg, sampler, evaluator, **kwargs)
        self.reducer = GradAllReducer()
        if hasattr(self.alg, "set_lockstep_replicas"):   # every rank computes gradient k together: the precision guards may use a collective
            self.alg.set_lockstep_replicas(True)
        if next(self.networks.parameters()).is_cuda or world_size() == 1 or not torch.cuda.is_available():
            broadcast_parameters(self.networks, src=0)
        else:   # replicas must live on their GPU before the first RCCL collective
            self.networks.to(torch.device("cuda", torch.cuda.current_device()))
            broadcast_parameters(self.networks, src=0)
        self.is_chief = rank() == 0
        if not self.is_chief:   # only rank 0 logs, saves and evaluates
            self.writer = None
            self.evaluator = None

    def step(self):
        samples = self._sample()
        self.networks.train()
        # no host sync between the backward sweep and the collective: the algorithms leave their loss scalars on the
        # device (read at log time), and the 1/N is folded into the Adam kernel when the algorithm supports it
        alg_tb_dict, update_info = self._gradient(samples)
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
