"""pyth_veh3dofconti_surrcstr_penalty model: veh3dofconti tracking next to ONE surrounding vehicle where the collision
constraint enters the stage reward as a smooth penalty, -15 (tanh(max(8 - 16 dis, 0) - 4) + 1) with dis = min circle
distance - 2 r of the CURRENT pose, so that plain FHADP trains on it.  The appended observation is the next surrounding
vehicle in the ego frame of the current state, the model never reports done, and `info["constraint"]` is the constraint
of the current pose.  Reference: gops/env/env_ocp/env_model/pyth_veh3dofconti_surrcstr_penalty_model.py:42-262
(example_train/fhadp/fhadp_mlp_veh3dofconti_surrcstr_penalty_serial.py).  Arithmetic: `surr_penalty` branches of the
GOPS_ENV_VEH3DOF_SURR kernels (csrc/env_models.h surr_penalty, rollout_fwd.hip, rollout_bwd.hip)."""
from gops_amd.env.env_ocp.env_model.pyth_veh3dofconti_surrcstr_model import Veh3dofcontiSurrCstrModel


class Veh3dofcontiSurrCstrPenaltyModel(Veh3dofcontiSurrCstrModel):
    n_constraint = 1
    # -(dx^2 + dy^2 + 0.1 dphi^2 + 0.1 du^2 + 0.5 v^2 + 0.5 omega^2 + 0.5 steer^2 + 0.5 a_x^2 + penalty)   (:159-169);
    # order of GopsEnv.reward_w: dx, dy, dphi, du, omega, steer, a_x, v
    reward_weights = (1.0, 1.0, 0.1, 0.1, 0.5, 0.5, 0.5, 0.5)

    def __init__(self, pre_horizon: int = 10, surr_veh_num: int = 1, **kwargs):
        if surr_veh_num != 1:
            raise RuntimeError("pyth_veh3dofconti_surrcstr_penalty: the reference reward (`punish_dis.squeeze()`, "
                               ":168) is only defined for surr_veh_num = 1")
        super().__init__(pre_horizon=pre_horizon, surr_veh_num=1, **kwargs)

    def hip_constants(self):
        c = super().hip_constants()
        c["surr"]["penalty"] = True
        return c


def env_model_creator(**kwargs):
    """make env model `pyth_veh3dofconti_surrcstr_penalty`"""
    return Veh3dofcontiSurrCstrPenaltyModel(**kwargs)
