"""pyth_veh2dofconti model: 2-DOF lateral vehicle dynamics at constant speed (u = 5 m/s) tracking one of the analytic
reference paths; state (y, phi, v, omega), action steer, obs = (y - y_ref0, phi - phi_ref0, v, omega, y - y_ref_1 ..
y - y_ref_P), info["ref_points"] [P+1, 2] = (y, phi) (reference:
gops/env/env_ocp/env_model/pyth_veh2dofconti_model.py:24-174, vehicle parameters gops/env/env_ocp/pyth_veh2dofconti.py:24-34
- the model of fhadp_mlp_veh2dofconti_serial.py / infadp_mlp_veh2dofconti_offserial.py).  Arithmetic in csrc/env_models.h
(veh2_f_xu, veh2_reward) and csrc/aux_kernels.hip (ref_point)."""
from typing import Dict, Optional, Union

import numpy as np
import torch

from gops_amd import hip_backend as hb
from gops_amd.env.env_ocp.env_model.pyth_base_model import PythBaseModel
from gops_amd.env.env_ocp.resources.ref_traj_params import ref_constants


class Veh2dofcontiModel(PythBaseModel):
    hip_kind = hb.ENV_VEH2DOF

    def __init__(self, pre_horizon: int = 10, device: Union[torch.device, str, None] = None,
                 path_para: Optional[Dict[str, Dict]] = None, u_para: Optional[Dict[str, Dict]] = None,
                 max_steer: float = np.pi / 6, **kwargs):
        # custom reference-trajectory parameters travel to the kernels as a table of folded constants (GopsEnv.ref_c)
        self.ref_c = ref_constants(path_para, u_para) if (path_para is not None or u_para is not None) else None
        self.pre_horizon = pre_horizon
        super().__init__(obs_dim=4 + pre_horizon, action_dim=1, dt=0.1, action_lower_bound=[-max_steer],
                         action_upper_bound=[max_steer], device=device)


def env_model_creator(**kwargs):
    """make env model `pyth_veh2dofconti`"""
    return Veh2dofcontiModel(**kwargs)
