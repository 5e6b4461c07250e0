"""Action distributions used by deterministic ADP policies.

The ADP algorithms on this path only ever use the Dirac distribution (the policy output IS the
action; reference gops/utils/act_distribution_type.py:141-150 and the mixin
gops/utils/act_distribution_cls.py:13-27 that samplers/evaluators call through
`networks.create_action_distributions(logits)`).
"""
import torch


class DiracDistribution:
    def __init__(self, logits: torch.Tensor):
        self.logits = logits

    def sample(self):
        return self.logits, torch.tensor([0.0])

    def mode(self):
        return self.logits


class Action_Distribution:
    """Mixin: builds `self.action_distribution_cls(logits)` and attaches the action limits."""

    def get_act_dist(self, logits):
        dist = getattr(self, "action_distribution_cls")(logits)
        if hasattr(self, "act_high_lim"):
            dist.act_high_lim = getattr(self, "act_high_lim")
            dist.act_low_lim = getattr(self, "act_low_lim")
        return dist
