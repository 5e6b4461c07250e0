"""OptController - receding-horizon optimal control over an environment model on the MI355X: single shooting and
direct collocation.

Interface of the reference's gops/sys_simulator/opt_controller.py:30-330 (`OptController(model, num_pred_step,
ctrl_interval, gamma, ..., mode="collocation")`, `controller(x, info) -> first optimal action`, warm start by shifting the
previous solution).  The reference rolls the raw model out step by step in Python (`_rollout`, :240-300) and gets the
Jacobian of the cost from autograd; here one cost + Jacobian evaluation is ONE forward and ONE backward launch of the
horizon-rollout kernels in their raw-action open-loop mode (`GopsRolloutDesc.open_loop = 2`: the decision variables are
the model's actions themselves, no tanh / ScaleAction; `GopsEnv.no_mask_at_done`: the raw model keeps stepping after its
done test fired, as in :261-265):

    cost(u_0 .. u_{T-1}) = - sum_i gamma^i r_i ,      d cost / d u   from  gops_rollout_backward_open_loop.

Box constraints on the actions are the model's action bounds.  The solver is scipy's L-BFGS-B (cyipopt, which the
reference calls, is not part of this environment).

`mode="collocation"` (the reference's default, :60, 77-82, 104-109): the decision variables are (action, state) at every
control point; the reference rolls ALL intervals out as one batch (`rollout_mode = "batch"`, :272-291: interval j starts
from the decision state of point j - 1, the first from x) and hands IPOPT the transition residuals
`true_state_j - decision_state_j = 0` as equality constraints (:196-215).  Here the batch of intervals is ONE open-loop
rollout launch (batch = control points, horizon = ctrl_interval), the cost gradient w.r.t. actions AND start states one
launch of `gops_rollout_backward_open_loop_adj` (ABI v8), and the transition Jacobian one forward + one backward launch over
obs_dim replicas of every interval, each seeded with one unit vector on its final state.  Solver: scipy's SLSQP with the
equality constraints and the box bounds on actions and states.  Collocation needs a model whose observation IS its state
and whose `forward` takes no `info` (pyth_lq, pyth_idpendulum, gym_cartpoleconti, gym_pendulum, pyth_mobilerobot) - for the others the
reference itself drops to its step-by-step rollout (:292-294); use `mode="shooting"` for them here.

Path constraints (`model.get_constraint`, :178-206: the reference evaluates it on all T + 1 states of the prediction and hands
IPOPT `-c >= 0` with a `jacrev` Jacobian): for the constrained vehicle models (pyth_veh3dofconti_surrcstr / _detour /
_errcstr, pyth_veh2dofconti_errcstr; shooting mode - they need `info`) the values are the per-step constraint output of ONE
rollout launch (`GopsRolloutOut.constraints`; the state the kernel does not emit - the initial one for the surrounding-vehicle
models - comes from `gops_env_constraint`), and the Jacobian is one forward + one backward launch over T n_constraint replicas
of the trajectory, replica (t, k) seeded with a unit d/d c_tk (`GopsRolloutIn.grad_constraint_step`).  Solver: SLSQP.

`use_terminal_cost` (:84-98, 312-317): `terminal_cost(state_T)` (the given torch function, else `model.get_terminal_cost` -
pyth_lq's x'Px) is evaluated with autograd on the device between the forward and the backward launch; its gradient seeds the
sweep through `grad_final_obs` (models whose observation is the state).

Not provided: the constraint of the surrcstr_penalty model (its info["constraint"] is computed on detached copies and carries
no gradient).
"""
import warnings
from typing import Dict, Optional

import numpy as np
import scipy.optimize as opt
import torch

from gops_amd import hip_backend as hb

_INFO = ("state", "ref_points", "path_num", "u_num", "ref_time", "surr_state")
# models whose observation is the state: adjoint I/O around the rollout (gops_rollout_backward_open_loop_adj)
_STATE_OBS_KINDS = (hb.ENV_LQ, hb.ENV_IDP, hb.ENV_CARTPOLE, hb.ENV_PENDULUM, hb.ENV_MOBILEROBOT)


class OptController:
    def __init__(self, model, num_pred_step: int, ctrl_interval: int = 1, gamma: float = 1.0,
                 use_terminal_cost: bool = False, terminal_cost=None, minimize_options: Optional[dict] = None,
                 verbose: int = 0, mode: str = "collocation", device=None):
        assert mode in ("shooting", "collocation")
        assert num_pred_step % ctrl_interval == 0, "ctrl_interval should be a factor of num_pred_step."
        base = model.unwrapped
        if mode == "collocation" and base.hip_kind not in _STATE_OBS_KINDS:
            # The batched collocation needs a model whose observation is its state and whose forward takes no info (pyth_lq,
            # pyth_idpendulum, gym_cartpoleconti, gym_pendulum, pyth_mobilerobot).  For the others the reference itself drops to
            # its step-by-step rollout (opt_controller.py:292-294); callers that rely on the default mode keep working here
            # through the shooting formulation: the transition equalities are eliminated instead of imposed - the same optimum
            # UNLESS state bounds are active: collocation hands obs_lower_bound / obs_upper_bound to the solver as variable bounds
            # (opt_controller.py:104-115), shooting has no state variables to bound and drops them.
            warnings.warn(f"OptController: mode='collocation' is not available for {type(base).__name__} (its forward needs `info`); "
                          "falling back to mode='shooting'.  The shooting formulation optimises the actions only: the model's "
                          "obs_lower_bound / obs_upper_bound, which collocation imposes as variable bounds on the states, are NOT "
                          "enforced - where they would be active the two modes find different optima (action bounds and the "
                          "model's inequality constraints are kept)")
            mode = "shooting"
        self.model, self.base = model, base
        self.terminal_cost = None
        if use_terminal_cost:   # (:84-98) the given function, else the model's own
            self.terminal_cost = terminal_cost if terminal_cost is not None else getattr(model, "get_terminal_cost", None)
            assert self.terminal_cost is not None, "Choose to use terminal cost, but there is no available terminal cost function."
            if base.hip_kind not in _STATE_OBS_KINDS:
                raise NotImplementedError("a terminal cost needs the adjoint of the final observation, which the kernels provide for "
                                          "the models whose observation is the state")
        elif terminal_cost is not None:
            warnings.warn("Choose not to use terminal cost, but a terminal cost function is given. This will be ignored.")
        self.obs_dim, self.action_dim, self.sim_dt = base.obs_dim, base.action_dim, base.dt
        self.gamma, self.ctrl_interval, self.num_pred_step = gamma, ctrl_interval, num_pred_step
        self.num_ctrl_points = num_pred_step // ctrl_interval
        self.mode, self.rollout_mode = mode, "kernel"
        self.minimize_options = dict(minimize_options or {})
        self.verbose = verbose
        lo = base.action_lower_bound.cpu().numpy().astype(np.float64)
        hi = base.action_upper_bound.cpu().numpy().astype(np.float64)
        if mode == "collocation":
            lo = np.concatenate((lo, base.obs_lower_bound.cpu().numpy().astype(np.float64)))
            hi = np.concatenate((hi, base.obs_upper_bound.cpu().numpy().astype(np.float64)))
            self.optimize_dim = self.action_dim + self.obs_dim
        else:
            self.optimize_dim = self.action_dim
        self.bounds = opt.Bounds(np.tile(lo, self.num_ctrl_points), np.tile(hi, self.num_ctrl_points))
        self.initial_guess = np.zeros(self.optimize_dim * self.num_ctrl_points)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        # the RAW model: identity action scaling, no observation clip, no MaskAtDone
        env = hb.make_env(base.hip_kind, base.obs_dim, base.action_dim, act_low=base.action_lower_bound.cpu(),
                          act_high=base.action_upper_bound.cpu(), min_action=base.action_lower_bound.cpu(),
                          max_action=base.action_upper_bound.cpu(), pre_horizon=getattr(base, "pre_horizon", 0),
                          **base.hip_constants())
        env.no_mask_at_done = 1
        # path constraints: the models that define get_constraint (all of them need `info`: shooting mode)
        self._cstr = getattr(base, "get_constraint", None) is not None and hb.has_constraints(env)
        if self._cstr:
            if env.surr_penalty:
                raise NotImplementedError("the constraint of pyth_veh3dofconti_surrcstr_penalty carries no gradient (the model computes "
                                          "it on detached copies): its OptController constraint is not provided")
            T, nc = num_pred_step, env.n_constraint
            self._nc = nc
            # errcstr models emit c(obs_t) at step t (t < H): one padded step more gives c(obs_T); the surrounding-vehicle
            # models emit c(state_{t+1}) and the initial state's value comes from gops_env_constraint
            self._cstr_err = bool(env.cstr_err)
            Hc = T + 1 if self._cstr_err else T
            mkc = lambda batch: hb.Rollout(env, None, batch=batch, horizon=Hc, gamma=gamma, finite_horizon=False, need_grad=True,
                                           device=self.device, raw_actions=True)
            self._cstr_ro, self._cstr_jac_ro = mkc(1), mkc(T * nc)
            seed = torch.zeros(Hc, T * nc, nc, dtype=torch.float32, device=self.device)
            for t in range(1, T + 1):       # row (t, k) of the Jacobian <- replica (t - 1) nc + k
                for k in range(nc):
                    seed[t if self._cstr_err else t - 1, (t - 1) * nc + k, k] = 1.0
            self._cstr_seed = seed
            self._cstr_zero = torch.zeros(T * nc, dtype=torch.float32, device=self.device)
            self._cstr_env = env
        self._rollout_obj = hb.Rollout(env, None, batch=1, horizon=num_pred_step, gamma=gamma, finite_horizon=False,
                                       need_grad=True, device=self.device, raw_actions=True)
        self._minus_one = torch.full((1,), -1.0, dtype=torch.float32, device=self.device)
        if mode == "collocation":
            n, ci, O = self.num_ctrl_points, ctrl_interval, self.obs_dim
            mk = lambda batch: hb.Rollout(env, None, batch=batch, horizon=ci, gamma=gamma, finite_horizon=False, need_grad=True,
                                          device=self.device, raw_actions=True)
            self._col, self._col_jac = mk(n), mk(n * O)
            # interval j covers the global steps j ci .. j ci + ci - 1: its discounted return enters the cost with gamma^(j ci)
            self._col_w = torch.tensor([-(gamma ** (j * ci)) for j in range(n)], dtype=torch.float32, device=self.device)
            self._zeros_nO = torch.zeros(n * O, dtype=torch.float32, device=self.device)
            self._eye_rep = torch.eye(O, dtype=torch.float32, device=self.device).repeat(n, 1).contiguous()   # row j O + k = e_k
        self._reset_statistics()

    # ---- collocation: (action, state) per control point ------------------------------------------
    def _col_split(self, inputs: np.ndarray, x):
        """-> (start states [n, O], held actions [n, ci, A]) of the decision vector `inputs` and the current state x."""
        n, A, O, ci = self.num_ctrl_points, self.action_dim, self.obs_dim, self.ctrl_interval
        z = torch.as_tensor(np.asarray(inputs, dtype=np.float32), device=self.device).reshape(n, A + O)
        x0 = torch.as_tensor(np.asarray(x), dtype=torch.float32, device=self.device).reshape(1, O)
        starts = torch.cat((x0, z[:-1, A:]), 0).contiguous()
        acts = z[:, :A].unsqueeze(1).expand(n, ci, A).contiguous()
        return z, starts, acts

    def _col_data(self, starts: torch.Tensor) -> Dict:
        data = {"obs": starts, "done": torch.zeros(starts.shape[0], dtype=torch.float32, device=self.device)}
        if self.base.hip_kind == hb.ENV_MOBILEROBOT:   # a deterministic cost: the obstacle follows its expected motion
            data["noise"] = torch.zeros(self.ctrl_interval, starts.shape[0], 2, dtype=torch.float32, device=self.device)
        return data

    def _col_cost_and_jac(self, inputs: np.ndarray, x, info: Dict):
        n, A, O = self.num_ctrl_points, self.action_dim, self.obs_dim
        self.system_simulations += 1
        z, starts, acts = self._col_split(inputs, x)
        res = self._col.forward(self._col_data(starts), head_pre=acts, want_final=self.terminal_cost is not None)
        cost = (self._col_w * res["v_pi"]).sum()
        gfo = None
        if self.terminal_cost is not None:      # on the true final state of the last interval (:312-317)
            tc, g_last = self._terminal(res["final_obs"][-1])
            cost = cost + tc
            gfo = torch.zeros(n, O, dtype=torch.float32, device=self.device)
            gfo[-1] = g_last
        g_act, g_obs = self._col.backward_open_loop_adj(self._col_w, grad_final_obs=gfo)
        jac = torch.zeros(n, A + O, dtype=torch.float32, device=self.device)
        jac[:, :A] = g_act.sum(1)               # the action of a point is held over its interval
        jac[:-1, A:] = g_obs[1:]                # the state of point j starts interval j + 1
        return float(cost.item()), jac.reshape(-1).double().cpu().numpy()

    def _trans_constraint_fcn(self, inputs: np.ndarray, x, info: Dict) -> np.ndarray:
        """true_state_j - decision_state_j for every control point (opt_controller.py:196-215)."""
        n, A = self.num_ctrl_points, self.action_dim
        self.constraint_evaluations += 1
        z, starts, acts = self._col_split(inputs, x)
        res = self._col.forward(self._col_data(starts), head_pre=acts, want_final=True)
        return (res["final_obs"] - z[:, A:]).reshape(-1).double().cpu().numpy()

    def _trans_constraint_jac(self, inputs: np.ndarray, x, info: Dict) -> np.ndarray:
        """Jacobian [n O, n (A + O)] of the transition residuals: block (j, j) = d true_state_j / d action_j and - I on the
        decision state, block (j, j - 1) = d true_state_j / d start state.  One forward + one sweep over O replicas of each
        interval, replica k seeded with e_k on its final state."""
        n, A, O, ci = self.num_ctrl_points, self.action_dim, self.obs_dim, self.ctrl_interval
        z, starts, acts = self._col_split(inputs, x)
        rep = lambda t: t.repeat_interleave(O, dim=0).contiguous()
        self._col_jac.forward(self._col_data(rep(starts)), head_pre=rep(acts))
        g_act, g_obs = self._col_jac.backward_open_loop_adj(self._zeros_nO, grad_final_obs=self._eye_rep)
        g_act = g_act.sum(1).reshape(n, O, A).double().cpu().numpy()       # [j, k, a] = d state_j[k] / d action_j[a]
        g_obs = g_obs.reshape(n, O, O).double().cpu().numpy()              # [j, k, i] = d state_j[k] / d start_j[i]
        D = A + O
        J = np.zeros((n * O, n * D))
        for j in range(n):
            J[j * O:(j + 1) * O, j * D:j * D + A] = g_act[j]
            J[j * O:(j + 1) * O, j * D + A:(j + 1) * D] = -np.eye(O)
            if j > 0:
                J[j * O:(j + 1) * O, (j - 1) * D + A:j * D] = g_obs[j]
        return J

    # ------------------------------------------------------------------------------------------
    def _batch(self, x, info: Dict):
        data = {"obs": torch.as_tensor(np.asarray(x), dtype=torch.float32, device=self.device).reshape(1, -1).contiguous(),
                "done": torch.zeros(1, dtype=torch.float32, device=self.device)}
        if self.base.hip_kind == hb.ENV_MOBILEROBOT:   # a deterministic cost: the obstacle follows its expected motion
            data["noise"] = torch.zeros(self.num_pred_step, 1, 2, dtype=torch.float32, device=self.device)
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
        res = self._rollout_obj.forward(self._batch(x, info), head_pre=self._actions(inputs), want_final=self.terminal_cost is not None)
        cost = -res["v_pi"][0]
        if self.terminal_cost is not None:   # + gamma^T terminal_cost(state_T) (:312-317), its gradient seeds the sweep
            tc, gfo = self._terminal(res["final_obs"][0])
            cost = cost + tc
            g, _ = self._rollout_obj.backward_open_loop_adj(self._minus_one, grad_final_obs=gfo.reshape(1, -1).contiguous())
        else:
            g = self._rollout_obj.backward_open_loop(self._minus_one)    # d(-v_pi)/d(action) [1, T, A]
        jac = g.reshape(self.num_ctrl_points, self.ctrl_interval, self.action_dim).sum(1).reshape(-1)
        return float(cost.item()), jac.double().cpu().numpy()

    def _terminal(self, final_obs: torch.Tensor):
        """(gamma^T terminal_cost(state_T), its gradient w.r.t. state_T): the user's / model's torch function, evaluated with
        autograd on the device between the forward and the backward launch."""
        xT = final_obs.detach().clone().requires_grad_(True)
        tc = self.terminal_cost(xT) * (self.gamma ** self.num_pred_step)
        (g,) = torch.autograd.grad(tc, xT)
        return tc.detach(), g.to(torch.float32)

    # ---- path constraints (shooting): -get_constraint(state_t, info_t) >= 0 for t = 0 .. T -----------------------
    def _cstr_actions(self, inputs: np.ndarray) -> torch.Tensor:
        u = self._actions(inputs)
        return torch.cat((u, u[:, -1:]), 1).contiguous() if self._cstr_err else u   # (errcstr: one padded step, see __init__)

    def _constraint_fcn(self, inputs: np.ndarray, x, info: Dict) -> np.ndarray:
        """[(T + 1) n_constraint]: -c of every state of the prediction, t-major (opt_controller.py:178-198)."""
        self.constraint_evaluations += 1
        data = self._batch(x, info)
        res = self._cstr_ro.forward(data, head_pre=self._cstr_actions(inputs), want_constraints=True)
        c = res["constraints"][:, 0]                                             # [Hc, nc]
        if not self._cstr_err:
            c0 = hb.env_constraint(self._cstr_env, data["obs"], {k: data[k] for k in ("state", "surr_state") if k in data})
            c = torch.cat((c0, c), 0)
        return (-c).reshape(-1).double().cpu().numpy()

    def _constraint_jac(self, inputs: np.ndarray, x, info: Dict) -> np.ndarray:
        """[(T + 1) n_constraint, n A]: one forward + one sweep over T n_constraint replicas, each seeded with one unit
        d/d c_tk; the rows of the initial state are zero."""
        T, nc, n, A = self.num_pred_step, self._nc, self.num_ctrl_points, self.action_dim
        R = T * nc
        data = {k: v.expand((R,) + tuple(v.shape[1:])).contiguous() for k, v in self._batch(x, info).items()}
        acts = self._cstr_actions(inputs)
        self._cstr_jac_ro.forward(data, head_pre=acts.expand((R,) + tuple(acts.shape[1:])).contiguous())
        g = self._cstr_jac_ro.backward_open_loop(self._cstr_zero, grad_constraint_step=self._cstr_seed)   # [R, Hc, A]
        g = g[:, :T].reshape(R, n, self.ctrl_interval, A).sum(2).reshape(R, n * A)
        J = np.zeros(((T + 1) * nc, n * A))
        J[nc:] = -g.double().cpu().numpy()
        return J

    def __call__(self, x: np.ndarray, info: Optional[Dict] = None) -> np.ndarray:
        """Optimal control input for the current state `x` (and model info, e.g. veh3dofconti's reference window)."""
        info = info or {}
        if self.mode == "collocation":
            res = opt.minimize(self._col_cost_and_jac, self.initial_guess, args=(x, info), jac=True, bounds=self.bounds,
                               method="SLSQP", options=self.minimize_options or {"maxiter": 200, "ftol": 1e-9},
                               constraints=[{"type": "eq", "fun": self._trans_constraint_fcn, "jac": self._trans_constraint_jac,
                                             "args": (x, info)}])
        elif self._cstr:
            res = opt.minimize(self._cost_fcn_and_jac, self.initial_guess, args=(x, info), jac=True, bounds=self.bounds,
                               method="SLSQP", options=self.minimize_options or {"maxiter": 200, "ftol": 1e-9},
                               constraints=[{"type": "ineq", "fun": self._constraint_fcn, "jac": self._constraint_jac,
                                             "args": (x, info)}])
        else:
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
