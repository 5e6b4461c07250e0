"""OptController - receding-horizon optimal control over an environment model, by single shooting on the MI355X.

Interface of the reference's gops/sys_simulator/opt_controller.py:30-330 (`OptController(model, num_pred_step,
ctrl_interval, gamma, ..., mode="shooting")`, `controller(x, info) -> first optimal action`, warm start by shifting the
previous solution).  The reference rolls the raw model out step by step in Python (`_rollout`, :240-300) and gets the
Jacobian of the cost from autograd; here one cost + Jacobian evaluation is ONE forward and ONE backward launch of the
horizon-rollout kernels in their raw-action open-loop mode (`GopsRolloutDesc.open_loop = 2`: the decision variables are
the model's actions themselves, no tanh / ScaleAction; `GopsEnv.no_mask_at_done`: the raw model keeps stepping after its
done test fired, as in :261-265):

    cost(u_0 .. u_{T-1}) = - sum_i gamma^i r_i ,      d cost / d u   from  gops_rollout_backward_open_loop.

Box constraints on the actions are the model's action bounds.  The solver is scipy's L-BFGS-B (cyipopt, which the
reference calls, is not part of this environment).  Not provided: `mode="collocation"` and path constraints
(`model.get_constraint`), which need IPOPT's general constraint handling, and user terminal-cost callbacks (a Python
function cannot run inside the kernel).
"""
from typing import Dict, Optional

import numpy as np
import scipy.optimize as opt
import torch

from gops_amd import hip_backend as hb

_INFO = ("state", "ref_points", "path_num", "u_num", "ref_time")


class OptController:
    def __init__(self, model, num_pred_step: int, ctrl_interval: int = 1, gamma: float = 1.0,
                 use_terminal_cost: bool = False, terminal_cost=None, minimize_options: Optional[dict] = None,
                 verbose: int = 0, mode: str = "shooting", device=None):
        if mode != "shooting":
            raise NotImplementedError("OptController on the HIP rollout supports mode='shooting' only")
        if use_terminal_cost or terminal_cost is not None:
            raise NotImplementedError("terminal-cost callbacks cannot run inside the rollout kernel")
        assert num_pred_step % ctrl_interval == 0, "ctrl_interval should be a factor of num_pred_step."
        base = model.unwrapped
        if base.hip_kind == hb.ENV_VEH_SURR:
            raise NotImplementedError("models with path constraints need a constrained solver (not provided)")
        self.model, self.base = model, base
        self.obs_dim, self.action_dim, self.sim_dt = base.obs_dim, base.action_dim, base.dt
        self.gamma, self.ctrl_interval, self.num_pred_step = gamma, ctrl_interval, num_pred_step
        self.num_ctrl_points = num_pred_step // ctrl_interval
        self.mode, self.rollout_mode, self.optimize_dim = mode, "kernel", self.action_dim
        self.minimize_options = dict(minimize_options or {})
        self.verbose = verbose
        lo = base.action_lower_bound.cpu().numpy().astype(np.float64)
        hi = base.action_upper_bound.cpu().numpy().astype(np.float64)
        self.bounds = opt.Bounds(np.tile(lo, self.num_ctrl_points), np.tile(hi, self.num_ctrl_points))
        self.initial_guess = np.zeros(self.optimize_dim * self.num_ctrl_points)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        # the RAW model: identity action scaling, no observation clip, no MaskAtDone
        env = hb.make_env(base.hip_kind, base.obs_dim, base.action_dim, act_low=base.action_lower_bound.cpu(),
                          act_high=base.action_upper_bound.cpu(), min_action=base.action_lower_bound.cpu(),
                          max_action=base.action_upper_bound.cpu(), pre_horizon=getattr(base, "pre_horizon", 0),
                          **base.hip_constants())
        env.no_mask_at_done = 1
        self._rollout_obj = hb.Rollout(env, None, batch=1, horizon=num_pred_step, gamma=gamma, finite_horizon=False,
                                       need_grad=True, device=self.device, raw_actions=True)
        self._minus_one = torch.full((1,), -1.0, dtype=torch.float32, device=self.device)
        self._reset_statistics()

    # ------------------------------------------------------------------------------------------
    def _batch(self, x, info: Dict):
        data = {"obs": torch.as_tensor(np.asarray(x), dtype=torch.float32, device=self.device).reshape(1, -1).contiguous(),
                "done": torch.zeros(1, dtype=torch.float32, device=self.device)}
        for k in _INFO:
            if info and k in info:
                v = torch.as_tensor(np.asarray(info[k]), dtype=torch.float32, device=self.device)
                data[k] = v.reshape((1,) + tuple(v.shape)).contiguous()
        return data

    def _actions(self, inputs: np.ndarray) -> torch.Tensor:
        u = torch.as_tensor(np.asarray(inputs, dtype=np.float32), device=self.device)
        u = u.reshape(self.num_ctrl_points, self.action_dim).repeat_interleave(self.ctrl_interval, dim=0)
        return u.reshape(1, self.num_pred_step, self.action_dim).contiguous()

    def _rollout(self, inputs: np.ndarray, x, info: Dict):
        """-> (final observation [obs_dim], discounted stage costs [T]) of the action sequence `inputs`."""
        self.system_simulations += 1
        res = self._rollout_obj.forward(self._batch(x, info), head_pre=self._actions(inputs), want_rewards=True, want_final=True)
        gam = torch.tensor([self.gamma ** i for i in range(self.num_pred_step)], dtype=torch.float32, device=self.device)
        return res["final_obs"][0], -res["rewards"][:, 0] * gam

    def _cost_fcn_and_jac(self, inputs: np.ndarray, x, info: Dict):
        """Value and Jacobian of the cost: one forward and one backward launch."""
        self.system_simulations += 1
        res = self._rollout_obj.forward(self._batch(x, info), head_pre=self._actions(inputs))
        g = self._rollout_obj.backward_open_loop(self._minus_one)        # d(-v_pi)/d(action) [1, T, A]
        jac = g.reshape(self.num_ctrl_points, self.ctrl_interval, self.action_dim).sum(1).reshape(-1)
        return float(-res["v_pi"][0].item()), jac.double().cpu().numpy()

    def __call__(self, x: np.ndarray, info: Optional[Dict] = None) -> np.ndarray:
        """Optimal control input for the current state `x` (and model info, e.g. veh3dofconti's reference window)."""
        info = info or {}
        res = opt.minimize(self._cost_fcn_and_jac, self.initial_guess, args=(x, info), jac=True, bounds=self.bounds,
                           method="L-BFGS-B", options=self.minimize_options or None)
        self.last_result = res
        # warm start of the next call: drop the first control point, repeat the last (:158-160)
        self.initial_guess = np.concatenate((res.x[self.optimize_dim:], res.x[-self.optimize_dim:]))
        if self.verbose > 0:
            self._print_statistics(res)
        return res.x.reshape(self.num_ctrl_points, self.optimize_dim)[0, :self.action_dim]

    def reset(self):
        self.initial_guess = np.zeros(self.optimize_dim * self.num_ctrl_points)

    def _reset_statistics(self):
        self.constraint_evaluations, self.system_simulations = 0, 0

    def _print_statistics(self, res, reset=True):
        print(f"OptController: cost {res.fun:.6g}, {res.nit} iterations, {self.system_simulations} rollout evaluations, "
              f"success={res.success}")
        if reset:
            self._reset_statistics()
