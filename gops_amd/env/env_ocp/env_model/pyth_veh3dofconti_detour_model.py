"""pyth_veh3dofconti_detour model: the surrounding-vehicle model with ONE parked vehicle to drive around, three
constraints (vehicle distance, upper and lower road boundary on the ego circles) and detour-specific reward weights.
Reference: gops/env/env_ocp/env_model/pyth_veh3dofconti_detour_model.py:40-181."""
from gops_amd.env.env_ocp.env_model.pyth_veh3dofconti_surrcstr_model import Veh3dofcontiSurrCstrModel


class Veh3dofcontiDetourModel(Veh3dofcontiSurrCstrModel):
    n_constraint = 3
    # -0.01 * (10 dx^2 + 2 dy^2 + 500 dphi^2 + 5 du^2 + 1000 omega^2 + 1000 steer^2 + 50 a_x^2)   (:153-173)
    reward_weights = (0.1, 0.02, 5.0, 0.05, 10.0, 10.0, 0.5)
    lane_width = 4.0
    road_upper, road_lower = 0.5 * 4.0, -1.5 * 4.0

    def __init__(self, pre_horizon: int = 10, surr_veh_num: int = 1, **kwargs):
        super().__init__(pre_horizon=pre_horizon, surr_veh_num=surr_veh_num, **kwargs)


def env_model_creator(**kwargs):
    """make env model `pyth_veh3dofconti_detour`"""
    return Veh3dofcontiDetourModel(**kwargs)
