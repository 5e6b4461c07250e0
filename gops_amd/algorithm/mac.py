"""MAC - mixed actor critic - on the fused HIP rollout.

Reference: gops/algorithm/mac.py (ApproxContainer :29-84, MAC :87-274).  Its two losses are INFADP's: PEV regresses
V(o) onto `sum_t gamma^t reward_scale r_t + (~d) gamma^n V_target(o_n)` from a no-grad model rollout (:213-243), PIM
ascends the same quantity through policy, model and the target value's input (:245-274); Adam per network and a
Polyak update of the updated network's target (:150-163).  What the reference adds on top is the "iterative Bayes
estimator" of a model-bias term `delta` (:170-205): `compute_loss_v` draws it (`np.random.multivariate_normal`) but
`dynamic_model_forward` replaces any non-None `delta` by zeros before every model step (:165-168), so the sampled
correction never reaches a loss - the gradients are exactly INFADP's with `reward_scale` (1) on the rewards.  This class
therefore IS the INFADP implementation (same kernels, HIP-graph replay, device Adam) under MAC's registry name and
attribute surface; the estimator's host-side sampling, which has no effect on any output, is not reproduced (it
consumes numpy RNG state in the reference).  The reference passes an empty info dict to the model (its shipped
examples use the gym-style cartpole / pendulum models); here the batch's info rides along, so the reference-trajectory
models work too.
"""
__all__ = ["MAC"]

from gops_amd.algorithm.infadp import INFADP, ApproxContainer  # noqa: F401  (create_alg looks the container up here)


class MAC(INFADP):
    """gamma, tau, pev_step, pim_step, forward_step as in the reference (mac.py:100-116)."""

    def __init__(self, index=0, **kwargs):
        super().__init__(index, **kwargs)
        self.reward_scale = 1   # mac.py:109; multiplies every model reward - at its only value the INFADP arithmetic
        self.delta = None       # mac.py:111 (see the module docstring)

    @property
    def adjustable_parameters(self):
        return ("gamma", "tau", "pev_step", "pim_step", "forward_step")
