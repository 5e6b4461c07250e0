"""pyth_mobilerobot model: a differential-drive robot tracking the line y = 0 at 0.3 m/s while avoiding one moving obstacle
robot; state = observation [13] = ego (x, y, theta, v, w), tracking errors (e_y, e_theta, e_v), obstacle (x, y, theta, v, w);
action (v_cmd, w_cmd); dt = 0.2; info["constraint"] [B, 1] = 0.89 - distance(obstacle, ego) of the new state (reference:
gops/env/env_ocp/env_model/pyth_mobilerobot_model.py:24-213 - the model of example_train/spil/spil_mlp_mobilerobot_
{offserial,async}.py).  Arithmetic in csrc/env_models.h (mob_forward / mob_backward).

The obstacle moves with noise: every model step the reference draws np.random.normal(0, 0.03) / (0, 0.02) per trajectory
(:141-167).  Here the draws of a whole rollout are one device tensor [H, B, 2] (`hip_backend.mobilerobot_noise`,
GopsRolloutIn.noise); a caller can hand in its own through data["noise"] / info["noise"] (the parity tests replay the
reference's draws that way)."""
from typing import Any, Dict, Union

import numpy as np
import torch

from gops_amd import hip_backend as hb
from gops_amd.env.env_ocp.env_model.pyth_base_model import PythBaseModel


class PythMobilerobotModel(PythBaseModel):
    hip_kind = hb.ENV_MOBILEROBOT

    def __init__(self, device: Union[torch.device, str, None] = None, **kwargs: Any):
        self.n_obstacle = 1
        self.safe_margin = 0.15
        robot = [30.0, 30.0, 2 * np.pi, 1.0, np.pi / 2]
        lb_state = [-30.0, -30.0, -2 * np.pi, -1.0, -np.pi / 2] + [-30.0, -np.pi, -2.0] + [-v for v in robot] * self.n_obstacle
        hb_state = [60.0, 30.0, 2 * np.pi, 1.0, np.pi / 2] + [30.0, np.pi, 2.0] + robot * self.n_obstacle
        self.state_dim = len(lb_state)
        super().__init__(obs_dim=len(lb_state), action_dim=2, dt=0.2, obs_lower_bound=lb_state, obs_upper_bound=hb_state,
                         action_lower_bound=[-0.4, -np.pi / 3], action_upper_bound=[0.4, np.pi / 3], device=device)

    def hip_constants(self) -> Dict:
        return dict(n_constraint=self.n_obstacle)


def env_model_creator(**kwargs):
    """make env model `pyth_mobilerobot`"""
    return PythMobilerobotModel(kwargs.get("device", None))
