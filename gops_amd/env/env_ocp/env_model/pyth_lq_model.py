"""pyth_lq model: x' = (I - A dt)^-1 (x + dt B u), r = scale * (shift - (x'Qx + u'Ru)) on the
current x, never done (reference: gops/env/env_ocp/resources/lq_base.py:35-141,317-357;
creator gops/env/env_ocp/env_model/pyth_lq_model.py:18-34)."""
from typing import Dict, Union

import numpy as np
import torch

from gops_amd import hip_backend as hb
from gops_amd.env.env_ocp.env_model.pyth_base_model import PythBaseModel
from gops_amd.env.env_ocp.resources import lq_configs


class LQDynamics:
    def __init__(self, config: dict, device=None):
        f32 = dict(dtype=torch.float32, device=device)
        self.A, self.B = torch.as_tensor(config["A"], **f32), torch.as_tensor(config["B"], **f32)
        self.Q, self.R = torch.as_tensor(config["Q"], **f32), torch.as_tensor(config["R"], **f32)
        self.time_step = config["dt"]
        self.reward_scale, self.reward_shift = config["reward_scale"], config["reward_shift"]
        self.state_dim = self.A.shape[0]
        # (I - A dt)^-1 in fp32, as the reference forms it (lq_base.py:55-57), on the host
        ia = torch.eye(self.state_dim) - self.A.cpu() * self.time_step
        self.inv_IA = torch.linalg.pinv(ia).to(device)
        self.device = device
        self._KP = None

    def compute_control_matrix(self):
        """Discounted LQR gain and cost-to-go (lq_base.py:59-71); init-time only, float64."""
        from scipy.linalg import solve_discrete_are
        gamma = 0.99
        A0 = self.A.cpu().numpy().astype("float64")
        A = np.linalg.pinv(np.eye(A0.shape[0]) - A0 * self.time_step) * np.sqrt(gamma)
        B = A @ self.B.cpu().numpy().astype("float64") * self.time_step
        Q = np.diag(self.Q.cpu().numpy()).astype("float64")
        R = np.diag(self.R.cpu().numpy()).astype("float64")
        P = solve_discrete_are(A, B, Q, R)
        K = np.linalg.pinv(R + B.T @ P @ B) @ B.T @ P @ A
        return K, P

    @property
    def K(self):
        if self._KP is None:
            self._KP = self.compute_control_matrix()
        return self._KP[0]

    @property
    def P(self):
        if self._KP is None:
            self._KP = self.compute_control_matrix()
        return self._KP[1]


class LqModel(PythBaseModel):
    hip_kind = hb.ENV_LQ

    def __init__(self, config: dict, device: Union[torch.device, str, None] = None):
        lo, hi = np.array(config["state_low"]), np.array(config["state_high"])
        alo, ahi = np.array(config["action_low"]), np.array(config["action_high"])
        super().__init__(obs_dim=lo.shape[0], action_dim=alo.shape[0], dt=config["dt"],
                         obs_lower_bound=lo, obs_upper_bound=hi, action_lower_bound=alo,
                         action_upper_bound=ahi, device=device)
        self.dynamics = LQDynamics(config, device)

    def hip_constants(self) -> Dict:
        d = self.dynamics
        return dict(lq=dict(inv_IA=d.inv_IA.cpu(), B=d.B.cpu(), Q=d.Q.cpu(), R=d.R.cpu(), dt=d.time_step,
                            reward_scale=d.reward_scale, reward_shift=d.reward_shift))

    def get_terminal_cost(self, obs: torch.Tensor) -> torch.Tensor:
        P = torch.as_tensor(self.dynamics.P, dtype=torch.float32, device=obs.device)
        return obs @ P @ (obs.T if obs.dim() == 2 else obs)   # (lq_base.py:356-357; .T of a 1-D tensor is the tensor itself)


def env_model_creator(**kwargs):
    """make env model `pyth_lq`; `lq_config` is a name ("s4a2"), a dict, or None (-> s3a1)."""
    lqc = kwargs.get("lq_config", None)
    if lqc is None:
        config = lq_configs.config_s3a1
    elif isinstance(lqc, str):
        assert hasattr(lq_configs, "config_" + lqc)
        config = getattr(lq_configs, "config_" + lqc)
    elif isinstance(lqc, dict):
        config = lqc
    else:
        raise RuntimeError("lq_config invalid")
    return LqModel(config, kwargs.get("device", None))
