"""Prioritized replay buffer whose storage AND priority trees live in HBM.

Interface and arithmetic of the reference's gops/trainer/buffer/prioritized_replay_buffer.py:22-151 (proportional
prioritization: alpha = 0.6, beta 0.4 -> 1 in steps of 0.01 per sampled batch, epsilon = 1e-6, new transitions enter at the
running maximum priority; `sample_batch` adds "idx" (tree index of the leaf, int32) and "weight" (importance weight divided by the
largest possible one) to the batch; `update_batch(idxes, priorities)` sets (|priority| + eps)^alpha).

The reference walks a numpy sum / min tree in Python: one root-to-leaf descent per sampled transition (`get_leaf`, :84-99) and a
set-based bottom-up repair per update (:131-151).  Here both trees are device tensors in the same array layout (2 N - 1 nodes,
leaf of slot i at i + N - 1, children of p at 2 p + 1 / 2 p + 2) and every operation is vectorised over the batch:
  * sampling: one stratified uniform draw per segment of the total (:104-110), then ALL descents together - ceil(log2(N)) + 1
    rounds of gather / compare / select on [batch] tensors, no host synchronisation;
  * storing / updating: the touched leaves are written, then each tree level above them is recomputed from its children for the
    (deduplicated) parents of the level below - ceil(log2(N)) + 1 rounds.
Trees are float64 like numpy's.  The tree has the reference's shape for ANY `buffer_max_size` (not only powers of two): leaves
sit on the last two levels, so a descent stops as soon as the node it reached has no children.
"""
import math

import numpy as np
import torch

from gops_amd.trainer.buffer.replay_buffer import ReplayBuffer

__all__ = ["PrioritizedReplayBuffer"]


class PrioritizedReplayBuffer(ReplayBuffer):
    def __init__(self, index=0, **kwargs):
        super().__init__(index, **kwargs)
        n = self.max_size
        self.sum_tree = torch.zeros(2 * n - 1, dtype=torch.float64, device=self.device)
        self.min_tree = torch.full((2 * n - 1,), float("inf"), dtype=torch.float64, device=self.device)
        self.alpha, self.beta, self.beta_increment, self.epsilon = 0.6, 0.4, 0.01, 1e-6
        self.max_priority = torch.ones((), dtype=torch.float64, device=self.device)   # 1.0 ** alpha (device scalar: no sync on update)
        self._depth = int(math.ceil(math.log2(max(2 * n - 1, 2)))) + 1                # rounds that cover every root-to-leaf path

    # ---- tree maintenance -----------------------------------------------------------------------------------------
    def _repair(self, tree_idx: torch.Tensor):
        """Recompute every ancestor of the given nodes (reference: update_tree :73-82 / the lazy loop :138-151): a FIXED number of
        rounds, each recomputing the parents of the previous round's nodes from their children - no `unique`, no emptiness test,
        no boolean mask, i.e. no device-to-host synchronisation.  Duplicated parents write the same value twice (a parent only
        depends on its children); nodes that have reached the root keep recomputing the root; leaves sit on the last two levels,
        so a parent can be recomputed before its deeper child has been - the round in which the deeper path passes recomputes it
        again, and after `_depth` rounds (>= the longest leaf-to-root path) every touched path is final."""
        if self.max_size == 1:
            return   # the root is the only leaf
        nodes = tree_idx
        for _ in range(self._depth):
            nodes = torch.clamp((nodes - 1) // 2, min=0)
            left, right = 2 * nodes + 1, 2 * nodes + 2
            self.sum_tree[nodes] = self.sum_tree[left] + self.sum_tree[right]
            self.min_tree[nodes] = torch.minimum(self.min_tree[left], self.min_tree[right])

    def _rows(self, n):
        rows = super()._rows(n)
        tree_idx = rows + (self.max_size - 1)
        self.sum_tree[tree_idx] = self.max_priority
        self.min_tree[tree_idx] = self.max_priority
        self._repair(tree_idx)
        return rows

    # ---- sampling -------------------------------------------------------------------------------------------------
    def get_leaf(self, values: torch.Tensor):
        """Vectorised `get_leaf` (:84-99): (tree index, priority) of the leaf each prefix-sum value falls into."""
        node = torch.zeros_like(values, dtype=torch.long)
        value = values.clone()
        last = self.sum_tree.numel() - 1
        for _ in range(self._depth):
            left = 2 * node + 1
            inner = left <= last                      # nodes that still have children
            lc = left.clamp(max=last)
            lsum = self.sum_tree[lc]
            go_left = value <= lsum
            nxt = torch.where(go_left, lc, (lc + 1).clamp(max=last))
            value = torch.where(inner & ~go_left, value - lsum, value)
            node = torch.where(inner, nxt, node)
        return node, self.sum_tree[node]

    def sample_batch(self, batch_size: int) -> dict:
        total = self.sum_tree[0]
        segment = total / batch_size
        self.beta = min(1.0, self.beta + self.beta_increment)
        min_prob = self.min_tree[0] / total
        max_weight = (min_prob * self.size) ** (-self.beta)
        u = torch.rand(batch_size, generator=self._gen, device=self.device, dtype=torch.float64)
        values = (torch.arange(batch_size, device=self.device, dtype=torch.float64) + u) * segment   # uniform(i seg, (i + 1) seg)
        idxes, priorities = self.get_leaf(values)
        probs = priorities / total
        weights = (probs * self.size) ** (-self.beta) / max_weight
        ptrs = idxes - (self.max_size - 1)
        batch = {k: v.index_select(0, ptrs) for k, v in self.buf.items()}
        batch["idx"] = idxes.to(torch.int32)
        batch["weight"] = weights.to(torch.float32)
        return batch

    def update_batch(self, idxes, priorities) -> None:
        idxes = torch.as_tensor(np.asarray(idxes) if not torch.is_tensor(idxes) else idxes).to(self.device).long().reshape(-1)
        pr = torch.as_tensor(np.asarray(priorities) if not torch.is_tensor(priorities) else priorities)
        pr = (pr.detach().to(self.device, torch.float64).reshape(-1) + self.epsilon) ** self.alpha
        self.max_priority = torch.maximum(self.max_priority, pr.max())
        # duplicated indices: numpy's fancy assignment keeps the LAST occurrence; a device scatter with duplicates is unordered, so
        # every occurrence writes the value of the index's last occurrence (position table in a persistent scratch: no `unique`,
        # no host synchronisation)
        pos = torch.arange(idxes.numel(), device=self.device)
        last = getattr(self, "_last_pos", None)
        if last is None:
            last = self._last_pos = torch.zeros(self.sum_tree.numel(), dtype=torch.long, device=self.device)
        last[idxes] = 0
        last.scatter_reduce_(0, idxes, pos, reduce="amax", include_self=True)
        winner = pr[last[idxes]]
        self.sum_tree[idxes] = winner
        self.min_tree[idxes] = winner
        self._repair(idxes)
