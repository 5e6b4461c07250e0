"""pyth_veh3dofconti_errcstr model: pyth_veh3dofconti with two constraints on the tracking errors of the CURRENT
observation, info["constraint"] = (|delta_y| - y_error_tol, |delta_u| - u_error_tol) (reference:
gops/env/env_ocp/env_model/pyth_veh3dofconti_errcstr_model.py:21-55 - the model of
example_train/spil/spil_mlp_veh3dofconti_errcstr_offserial.py).  Runs on the GOPS_ENV_VEH3DOF_SURR kernels with no
surrounding vehicle (`cstr_err`): same constraint sums / products / adjoints as the other constrained models."""
from typing import Dict, Optional, Union

import numpy as np
import torch

from gops_amd import hip_backend as hb
from gops_amd.env.env_ocp.env_model.pyth_base_model import PythBaseModel
from gops_amd.env.env_ocp.resources.ref_traj_params import ref_constants
from gops_amd.env.env_ocp.env_model.pyth_veh3dofconti_surrcstr_model import TRACKING_WEIGHTS


class Veh3dofcontiErrCstrModel(PythBaseModel):
    hip_kind = hb.ENV_VEH_SURR

    def __init__(self, pre_horizon: int = 10, device: Union[torch.device, str, None] = None,
                 path_para: Optional[Dict[str, Dict]] = None, u_para: Optional[Dict[str, Dict]] = None,
                 y_error_tol: float = 0.2, u_error_tol: float = 2.0, **kwargs):
        # custom reference-trajectory parameters travel to the kernels as a table of folded constants (GopsEnv.ref_c)
        self.ref_c = ref_constants(path_para, u_para) if (path_para is not None or u_para is not None) else None
        self.pre_horizon = pre_horizon
        self.y_error_tol, self.u_error_tol = y_error_tol, u_error_tol
        super().__init__(obs_dim=6 + 4 * pre_horizon, action_dim=2, dt=0.1,
                         action_lower_bound=[-np.pi / 6, -3], action_upper_bound=[np.pi / 6, 3], device=device)

    def get_constraint(self, obs: torch.Tensor, info: Optional[Dict] = None) -> torch.Tensor:
        return self._hip_get_constraint(obs, info)

    def hip_constants(self) -> Dict:
        return dict(surr=dict(n_surr=0, n_constraint=2, veh_length=4.8, veh_width=2.0, reward_w=TRACKING_WEIGHTS,
                              err_tol=(self.y_error_tol, self.u_error_tol)))


def env_model_creator(**kwargs):
    """make env model `pyth_veh3dofconti_errcstr`"""
    return Veh3dofcontiErrCstrModel(**kwargs)
