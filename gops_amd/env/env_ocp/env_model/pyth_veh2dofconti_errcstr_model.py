"""pyth_veh2dofconti_errcstr model: pyth_veh2dofconti with one constraint on the lateral tracking error of the CURRENT
observation, info["constraint"] = |delta_y| - y_error_tol (reference:
gops/env/env_ocp/env_model/pyth_veh2dofconti_errcstr_model.py:18-49 - the model of
example_train/spil/spil_mlp_veh2dofconti_errcstr_offserial.py).  GOPS_ENV_VEH2DOF kernels with `cstr_err`."""
from typing import Dict, Optional

import torch

from gops_amd.env.env_ocp.env_model.pyth_veh2dofconti_model import Veh2dofcontiModel


class Veh2dofcontiErrCstrModel(Veh2dofcontiModel):
    def __init__(self, pre_horizon: int = 10, y_error_tol: float = 0.2, **kwargs):
        super().__init__(pre_horizon=pre_horizon, **kwargs)
        self.y_error_tol = y_error_tol

    def get_constraint(self, obs: torch.Tensor, info: Optional[Dict] = None) -> torch.Tensor:
        return self._hip_get_constraint(obs, info)

    def hip_constants(self) -> Dict:
        return dict(surr=dict(n_surr=0, n_constraint=1, veh_length=0.0, veh_width=0.0, reward_w=(0.0,) * 8,
                              err_tol=(self.y_error_tol, 0.0)))


def env_model_creator(**kwargs):
    """make env model `pyth_veh2dofconti_errcstr`"""
    return Veh2dofcontiErrCstrModel(**kwargs)
