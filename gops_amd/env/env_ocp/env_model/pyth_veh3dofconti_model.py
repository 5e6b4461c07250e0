"""pyth_veh3dofconti model: 3-DOF bicycle model tracking one of 4 analytic reference paths at one
of 2 speed profiles; obs = ego-frame errors of P+1 preview points (reference:
gops/env/env_ocp/env_model/pyth_veh3dofconti_model.py:64-210,
gops/env/env_ocp/resources/ref_traj_model.py).  Arithmetic in csrc/env_models.h (veh_f_xu,
veh_reward) and csrc/aux_kernels.hip (ref_point)."""
from typing import Dict, Optional, Union

import numpy as np
import torch

from gops_amd import hip_backend as hb
from gops_amd.env.env_ocp.env_model.pyth_base_model import PythBaseModel
from gops_amd.env.env_ocp.resources.ref_traj_params import ref_constants


class Veh3dofcontiModel(PythBaseModel):
    hip_kind = hb.ENV_VEH

    def __init__(self, pre_horizon: int = 10, device: Union[torch.device, str, None] = None,
                 path_para: Optional[Dict[str, Dict]] = None, u_para: Optional[Dict[str, Dict]] = None,
                 max_steer: float = np.pi / 6, **kwargs):
        # custom reference-trajectory parameters travel to the kernels as a table of folded constants (GopsEnv.ref_c)
        self.ref_c = ref_constants(path_para, u_para) if (path_para is not None or u_para is not None) else None
        self.pre_horizon = pre_horizon
        super().__init__(obs_dim=6 + 4 * pre_horizon, action_dim=2, dt=0.1,
                         action_lower_bound=[-max_steer, -3], action_upper_bound=[max_steer, 3],
                         device=device)


def env_model_creator(**kwargs):
    """make env model `pyth_veh3dofconti`"""
    return Veh3dofcontiModel(**kwargs)
