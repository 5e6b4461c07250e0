"""pyth_idpendulum model: inverted double pendulum on a cart, 5 explicit-Euler sub-steps of
2 ms per model step with force 500*a (reference:
gops/env/env_ocp/env_model/pyth_idpendulum_model.py:20-216).  The arithmetic lives in
csrc/env_models.h (idp_substep / idp_reward / idp_done)."""
from typing import Union

import numpy as np
import torch

from gops_amd import hip_backend as hb
from gops_amd.env.env_ocp.env_model.pyth_base_model import PythBaseModel


class PythInvertedpendulum(PythBaseModel):
    hip_kind = hb.ENV_IDP

    def __init__(self, device: Union[torch.device, str, None] = None):
        self.discrete_num = 5
        super().__init__(obs_dim=6, action_dim=1, dt=0.01, obs_lower_bound=[-np.inf] * 6,
                         obs_upper_bound=[np.inf] * 6, action_lower_bound=[-1.0],
                         action_upper_bound=[1.0], device=device)


def env_model_creator(**kwargs):
    """make env model `pyth_idpendulum`"""
    return PythInvertedpendulum(kwargs.get("device", None))
