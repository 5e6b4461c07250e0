"""gym_cartpoleconti model: cart-pole with a continuous force 10 a, explicit Euler step of 20 ms; done when the NEXT state
leaves |x| <= 2.4 or |theta| <= 12 degrees, reward = 1 - done (reference:
gops/env/env_gym/env_model/gym_cartpoleconti_model.py:24-129 - the model of the INFADP / MAC cartpoleconti example scripts).
The observation IS the state, there is no info.  Arithmetic: csrc/env_models.h (cart_forward / cart_backward)."""
import math
from typing import Union

import numpy as np
import torch

from gops_amd import hip_backend as hb
from gops_amd.env.env_ocp.env_model.pyth_base_model import PythBaseModel


class GymCartpolecontiModel(PythBaseModel):
    hip_kind = hb.ENV_CARTPOLE

    def __init__(self, device: Union[torch.device, str, None] = None, **kwargs):
        self.theta_threshold_radians = 12 * 2 * math.pi / 360
        self.x_threshold = 2.4
        fmax = float(np.finfo(np.float32).max)
        lb = [-self.x_threshold * 2, -fmax, -self.theta_threshold_radians * 2, -fmax]
        ub = [self.x_threshold * 2, fmax, self.theta_threshold_radians * 2, fmax]
        super().__init__(obs_dim=4, action_dim=1, dt=0.02, obs_lower_bound=lb, obs_upper_bound=ub,
                         action_lower_bound=[-1.0], action_upper_bound=[1.0], device=device)


def env_model_creator(**kwargs):
    """make env model `gym_cartpoleconti`"""
    return GymCartpolecontiModel(kwargs.get("device", None))
