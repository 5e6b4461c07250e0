"""MPG - mixed policy gradient - on the HIP kernels.

Same class surface as the reference's gops/algorithm/mpg.py (ApproxContainer :33-102, MPG :105-474): twin
action-value functions regressed onto the clipped double-Q backup (and, for `pge_method="mixed_state"`, a second
"model" pair), and a policy ascended along a MIX of the data-driven gradient d q1(o, pi(o)) and the model-driven
gradient of the H-step model return `sum_t gamma^t r_t + gamma^H q1_target(o_H, pi(o_H))` (:332-399).

Where the arithmetic runs: every network evaluation and backward is `gops_mlp_forward / _backward / _backward_x`
(the (obs, act) -> q networks over the concatenated input); the model rollout is the closed-loop
`gops_rollout_forward`, and its sweep is `gops_rollout_backward_adj`: seeded with the terminal term's
d/d(o_H), parameter gradients through step 0 only - the later steps act through `policy4rollout`, a frozen copy
of the same weights whose INPUT still carries gradient (:343-349).  Tanh squash / concatenation / loss scalars are
elementwise torch ops on the device.
"""
__all__ = ["MPG"]

import time
from copy import deepcopy
from typing import Tuple

import numpy as np
import torch

from gops_amd import hip_backend as hb
from gops_amd.algorithm.base import AlgorithmBase, ApprBase, batch_to_device, cuda_device_of, grad_buffers
from gops_amd.create_pkg.create_apprfunc import create_apprfunc
from gops_amd.create_pkg.create_env_model import create_env_model
from gops_amd.utils.common_utils import get_apprfunc_dict, make_adam
from gops_amd.utils.lazy_scalar import scalar
from gops_amd.utils.hip_graph import StepGraphCache
from gops_amd.utils.tensorboard_setup import tb_tags


class ApproxContainer(ApprBase):
    """q1, q2 (+ q1_model, q2_model for mixed_state), policy, policy4rollout and the frozen targets; construction
    order (and with it the RNG draws) follows mpg.py:43-63."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        pge_method = kwargs["pge_method"]
        q_args = get_apprfunc_dict("value", **kwargs)
        self.q1 = create_apprfunc(**q_args)
        self.q2 = create_apprfunc(**q_args)
        if pge_method == "mixed_state":
            self.q1_model = deepcopy(self.q1)
            self.q2_model = deepcopy(self.q2)
        policy_args = get_apprfunc_dict("policy", **kwargs)
        self.policy = create_apprfunc(**policy_args)
        create_apprfunc(**policy_args)   # the reference builds (and discards) a second policy here: same RNG stream
        self.policy4rollout = deepcopy(self.policy)
        self.q1_target = deepcopy(self.q1)
        self.q2_target = deepcopy(self.q2)
        if pge_method == "mixed_state":
            self.q1_model_target = deepcopy(self.q1_model)
            self.q2_model_target = deepcopy(self.q2_model)
        self.policy_target = deepcopy(self.policy)
        frozen = [self.policy4rollout, self.q1_target, self.q2_target, self.policy_target]
        if pge_method == "mixed_state":
            frozen += [self.q1_model_target, self.q2_model_target]
        for net in frozen:
            for p in net.parameters():
                p.requires_grad = False
        self.q1_optimizer = make_adam(self.q1.parameters(), lr=kwargs["value_learning_rate"])
        self.q2_optimizer = make_adam(self.q2.parameters(), lr=kwargs["value_learning_rate"])
        if pge_method == "mixed_state":
            self.q1_model_optimizer = make_adam(self.q1_model.parameters(), lr=kwargs["value_learning_rate"])
            self.q2_model_optimizer = make_adam(self.q2_model.parameters(), lr=kwargs["value_learning_rate"])
        self.policy_optimizer = make_adam(self.policy.parameters(), lr=kwargs["policy_learning_rate"])

    def create_action_distributions(self, logits):
        return self.policy.get_act_dist(logits)


class MPG(AlgorithmBase):
    """pge_method: "mixed_weight" (rule-based weights of the two gradients, eta / terminal_iter) or "mixed_state"
    (per-sample choice by the disagreement of the data and model backups, kappa); gamma; tau: Polyak factor;
    delay_update: policy update period; forward_step: model rollout length."""

    def __init__(self, index: int = 0, terminal_iter: int = 10000, eta: float = 0.1, kappa: float = 0.5,
                 gamma: float = 0.99, tau: float = 0.1, delay_update: int = 1, forward_step: int = 10, **kwargs):
        super().__init__(index, **kwargs)
        self.networks = ApproxContainer(**kwargs)
        self.envmodel = create_env_model(**kwargs)
        # the model return's sweep is gops_rollout_backward_adj: built for the models whose observation is the state
        kind = getattr(getattr(self.envmodel, "model", self.envmodel), "hip_kind", None)
        if kind is not None and kind not in (hb.ENV_LQ, hb.ENV_IDP, hb.ENV_CARTPOLE, hb.ENV_PENDULUM):
            raise RuntimeError(f"MPG on the HIP path supports pyth_lq, pyth_idpendulum, gym_cartpoleconti and gym_pendulum models "
                               f"(adjoint I/O of the rollout sweep); env model '{kwargs.get('env_id')}' is not among them")
        self.pge_method = kwargs["pge_method"]
        if self.pge_method == "mixed_weight":
            self.terminal_iter = terminal_iter
            self.eta = eta
        elif self.pge_method == "mixed_state":
            self.kappa = kappa
        else:
            raise AssertionError("the pge_method entry should be mixed_state or mixed_weight")
        self.gamma = gamma
        self.tau = tau
        self.reward_scale = 1.0
        self.delay_update = delay_update
        self.forward_step = forward_step
        self.tb_info = dict()
        self._cache = {}
        self._tmp = {}
        self._graphs = {}

    @property
    def adjustable_parameters(self):
        return ("gamma", "tau", "delay_update", "terminal_iter", "eta")

    # ---- update API --------------------------------------------------------------------------
    def local_update(self, data: dict, iteration: int) -> dict:
        # An update is ~60 small launches (five to nine network batches, one rollout, Adam per network, Polyak): at
        # replay batch sizes it is launch-bound, so gradient + Adam + target update replay as ONE HIP graph.
        start_time = time.time()
        batch = self._batch(data, iteration)
        step_policy = iteration % self.delay_update == 0
        opts = [getattr(self.networks, f"{n}_optimizer") for n in self._q_names() + (["policy"] if step_policy else [])]

        def update(b):
            scalars = self._gradient_kernels(b)
            self._update(iteration)
            return scalars

        cache = self._graphs.setdefault(step_policy, StepGraphCache())
        scalars = cache.run(self._signature(batch, step_policy), batch, update,
                            before_replay=lambda: [o.sync_hyper() for o in opts], on_replay=lambda: [o.advance() for o in opts],
                            work=batch["obs"].shape[0] * self.forward_step,
                            on_capture_fail=lambda: [o.resync_device_state() for o in opts])
        self._step_schedulers()
        return self._log(scalars, start_time)

    def _batch(self, data: dict, iteration: int):
        device = cuda_device_of(self.networks)
        batch = batch_to_device(data, device, ("obs", "act", "rew", "obs2", "done"))
        if self.pge_method == "mixed_weight":   # the iteration-dependent weights travel with the batch (graph input)
            batch["_mix"] = torch.tensor(self._weights(iteration), dtype=torch.float32).to(device, non_blocking=True)
        return batch

    def _scalar_names(self):
        names = []
        for suffix in [""] + (["_model"] if self.pge_method == "mixed_state" else []):
            names += [f"MPG/loss_q1{suffix}-RL iter", f"MPG/q1{suffix}_mean-RL iter", f"MPG/loss_q2{suffix}-RL iter",
                      f"MPG/q2{suffix}_mean-RL iter", f"MPG/loss_q{suffix}-RL iter"]
        names += ["MPG/data_w-RL iter", "MPG/model_w-RL iter"] if self.pge_method == "mixed_weight" else ["MPG/model_ratio-RL iter"]
        return names + ["MPG/data_loss-RL iter", "MPG/model_loss-RL iter", "MPG/loss_pi-RL iter"]

    def _log(self, scalars: torch.Tensor, start_time: float) -> dict:
        tb_info = {name: scalar(scalars, i) for i, name in enumerate(self._scalar_names())}   # (GOPS_EAGER_LOG=1: host sync here)
        tb_info[tb_tags["alg_time"]] = (time.time() - start_time) * 1000
        self.tb_info = tb_info
        return tb_info

    def _signature(self, batch, step_policy):
        nets = self.networks
        mods = [m for m in nets.children()]
        opts = [getattr(nets, f"{n}_optimizer") for n in self._q_names() + ["policy"]]
        return (step_policy, tuple((k, tuple(v.shape)) for k, v in batch.items()), self.forward_step, float(self.gamma),
                float(self.tau), float(self.reward_scale), float(getattr(self, "kappa", 0.0)),
                tuple((p.data_ptr(), 0 if p.grad is None else p.grad.data_ptr()) for m in mods for p in m.parameters()),
                tuple(o.storage_signature() for o in opts),
                tuple(sorted(obj.workspace.data_ptr() for obj in self._cache.values())),
                tuple(t.data_ptr() for bufs in self._tmp.values() for lst in bufs for t in lst))

    def _q_names(self):
        return ["q1", "q2"] + (["q1_model", "q2_model"] if self.pge_method == "mixed_state" else [])

    def get_remote_update_info(self, data: dict, iteration: int) -> Tuple[dict, dict]:
        tb_info = self._compute_gradient(data, iteration)
        update_info = {f"{n}_grad": [p.grad for p in getattr(self.networks, n).parameters()]
                       for n in self._q_names() + ["policy"]}
        update_info["iteration"] = iteration
        return tb_info, update_info

    def remote_update(self, update_info: dict):
        for n in self._q_names() + ["policy"]:
            for p, grad in zip(getattr(self.networks, n).parameters(), update_info[f"{n}_grad"]):
                p.grad = grad
        self._update(update_info["iteration"])
        self._step_schedulers()

    def _update(self, iteration):
        nets = self.networks
        for n in self._q_names():
            getattr(nets, f"{n}_optimizer").step()
        if iteration % self.delay_update == 0:
            nets.policy_optimizer.step()
        with torch.no_grad():
            torch._foreach_copy_(list(nets.policy4rollout.parameters()), list(nets.policy.parameters()))
            for n in self._q_names() + ["policy"]:   # p_targ <- (1 - tau) p_targ + tau p   (mpg.py:383-424)
                online = list(getattr(nets, n).parameters())
                target = list(getattr(nets, f"{n}_target").parameters())
                torch._foreach_mul_(target, 1 - self.tau)
                torch._foreach_add_(target, online, alpha=self.tau)

    # ---- kernels -----------------------------------------------------------------------------
    def _net(self, role: str, module, B: int, device) -> hb.MlpNet:
        """One MlpNet (own activation stash) per use of a network inside an update: a backward follows ITS forward."""
        key = (role, B, str(device))
        mlp = module.hip_mlp()
        net = self._cache.get(key)
        if net is None:
            net = self._cache[key] = hb.MlpNet(mlp, B, device=device)
        else:
            net.mlp = mlp
        return net

    def _rollout_for(self, B: int, device) -> hb.Rollout:
        nets = self.networks
        key = ("rollout", B, self.forward_step, float(self.gamma), str(device))
        pol = nets.policy.hip_mlp()
        ro = self._cache.get(key)
        if ro is None:
            env = self.envmodel.hip_env(nets.policy.act_low_lim.cpu().numpy(), nets.policy.act_high_lim.cpu().numpy())
            ro = self._cache[key] = hb.Rollout(env, pol, batch=B, horizon=self.forward_step, gamma=self.gamma,
                                               finite_horizon=False, need_grad=True, device=device)
        else:
            ro.set_policy(pol)
        return ro

    def _scratch_grads(self, module, tag: str):
        bufs = self._tmp.get(tag)
        layers = module.linear_layers()
        if bufs is None or bufs[0][0].device != layers[0].weight.device:
            bufs = self._tmp[tag] = ([torch.zeros_like(l.weight) for l in layers], [torch.zeros_like(l.bias) for l in layers])
        return bufs

    def _squash(self, pre):
        """(action, d action / d pre) of DetermPolicy's tanh squash (apprfunc/mlp.py: `_squash`)."""
        pol = self.networks.policy
        half = (pol.act_high_lim - pol.act_low_lim) / 2
        th = torch.tanh(pre)
        return half * th + (pol.act_high_lim + pol.act_low_lim) / 2, half * (1 - th * th)

    def _q_pair_gradient(self, names, o, a, r, o2, d, a2_targ, info):
        """Clipped double-Q regression of the pair `names` (mpg.py:236-289): fills their .grad, returns the backup."""
        nets = self.networks
        B, device = o.shape[0], o.device
        x = torch.cat([o, a], dim=-1)
        x2 = torch.cat([o2, a2_targ], dim=-1)
        q_t = [self._net(f"{n}_target@o2", getattr(nets, f"{n}_target"), B, device).forward(x2).squeeze(-1) for n in names]
        backup = r + self.gamma * (1 - d) * torch.min(q_t[0], q_t[1])
        loss = 0.0
        for n in names:
            net = self._net(f"{n}@data", getattr(nets, n), B, device)
            q = net.forward(x).squeeze(-1)
            diff = q - backup
            gw, gb = grad_buffers(getattr(nets, n))
            net.backward(x, ((2.0 / B) * diff).unsqueeze(-1).contiguous(), gw, gb)
            li = (diff * diff).mean()
            loss = loss + li
            info += [li, q.mean()]
        info.append(loss)
        return backup

    def _weights(self, iteration):
        """Rule-based weights of the data-driven and the model-driven gradient (mpg.py:292-314), float64 like there."""
        lam = np.clip(1.0 - self.eta + 2.0 * self.eta / self.terminal_iter * iteration, 0, 1.5)
        if lam < 1.0:
            biases = np.array([np.power(lam, i) for i in [0, self.forward_step]])
        else:
            biases = np.array([np.power(2 - lam, self.forward_step - i) for i in [0, self.forward_step]])
        ws = torch.softmax(torch.tensor(1.0 / (biases + 1e-8)), dim=0)
        return float(ws[0]), float(ws[1])

    def _compute_gradient(self, data: dict, iteration: int) -> dict:
        start_time = time.time()
        return self._log(self._gradient_kernels(self._batch(data, iteration)), start_time)

    def _gradient_kernels(self, batch) -> torch.Tensor:
        """Enqueue one compute_gradient (mpg.py:161-218); returns the logged scalars (order of `_scalar_names`) as one
        device tensor, without synchronising."""
        nets = self.networks
        o, a, o2, d = batch["obs"], batch["act"], batch["obs2"], batch["done"]
        device = o.device
        r = batch["rew"] * self.reward_scale
        B, O, H = o.shape[0], o.shape[1], self.forward_step
        info = []

        # ---- action-value regression ------------------------------------------------------------
        a2_targ, _ = self._squash(self._net("policy_target@o2", nets.policy_target, B, device).forward(o2))
        backup_data = self._q_pair_gradient(["q1", "q2"], o, a, r, o2, d, a2_targ, info)
        if self.pge_method == "mixed_state":
            backup_model = self._q_pair_gradient(["q1_model", "q2_model"], o, a, r, o2, d, a2_targ, info)

        # ---- per-sample weights of the two returns in the policy loss ------------------------------
        if self.pge_method == "mixed_weight":
            data_w, model_w = batch["_mix"][0], batch["_mix"][1]
            w_data, w_model = (data_w / B).expand(B), (model_w / B).expand(B)
        else:
            cond = (torch.abs(backup_data - backup_model) < self.kappa * backup_data.std()).float()
            w_model, w_data = cond / B, (1 - cond) / B

        # ---- data return q1(o, pi(o)) and its gradient into the policy (q parameters frozen) ------
        pol_o = self._net("policy@o", nets.policy, B, device)
        a0, da0 = self._squash(pol_o.forward(o))
        xq = torch.cat([o, a0], dim=-1)
        q_data_net = self._net("q1@pi", nets.q1, B, device)
        data_return = q_data_net.forward(xq).squeeze(-1)
        g_xq = q_data_net.backward_x(xq, (-w_data).unsqueeze(-1).contiguous())
        gw_d, gb_d = self._scratch_grads(nets.policy, "data")
        pol_o.backward(o, (g_xq[:, O:] * da0).contiguous(), gw_d, gb_d)

        # ---- model return: H closed-loop model steps + gamma^H q1_target(o_H, pi(o_H)) ----------
        ro = self._rollout_for(B, device)
        res = ro.forward({"obs": o, "done": torch.zeros(B, dtype=torch.float32, device=device)}, want_final=True)
        o_h = res["final_obs"]
        pol_h = self._net("policy@oH", nets.policy, B, device)
        a_h, da_h = self._squash(pol_h.forward(o_h))
        xq_h = torch.cat([o_h, a_h], dim=-1)
        q_tail_net = self._net("q1_target@oH", nets.q1_target, B, device)
        gamma_h = float(self.gamma ** H)
        model_return = self.reward_scale * res["v_pi"] + gamma_h * q_tail_net.forward(xq_h).squeeze(-1)
        g_xq_h = q_tail_net.backward_x(xq_h, (-gamma_h * w_model).unsqueeze(-1).contiguous())
        gw_t, gb_t = self._scratch_grads(nets.policy, "tail")
        g_oh = pol_h.backward_x(o_h, (g_xq_h[:, O:] * da_h).contiguous(), gw_t, gb_t) + g_xq_h[:, :O]
        gw, gb = grad_buffers(nets.policy)
        ro.backward_adj((-self.reward_scale * w_model).contiguous(), gw, gb, grad_final_obs=g_oh.contiguous(),
                        first_step_only=True)
        torch._foreach_add_(gw + gb, gw_d + gb_d)
        torch._foreach_add_(gw + gb, gw_t + gb_t)

        # ---- log ---------------------------------------------------------------------------------
        data_loss, model_loss = -data_return.mean(), -model_return.mean()
        if self.pge_method == "mixed_weight":
            info += [data_w, model_w]
            loss_pi = data_w * data_loss + model_w * model_loss
        else:
            info.append(cond.mean())
            loss_pi = -(w_model * model_return + w_data * data_return).sum()
        info += [data_loss, model_loss, loss_pi]
        return torch.stack([x.reshape(()) for x in info])
