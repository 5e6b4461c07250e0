"""Shared test helpers: build oracle-side envs / nets from a golden fixture."""
import numpy as np
import torch

from oracle import adp_oracle as orc

INFO_KEYS = ("state", "ref_points", "path_num", "u_num", "ref_time", "surr_state")


def oracle_env(cfg, extra, golden=None):
    """Oracle env for a fixture's config.  `golden`: the fixture - constants the reference derived
    with LAPACK when the fixture was recorded (fp32 `pinv`, whose last bits are host dependent and
    get amplified by long unstable LQ horizons) are taken from it instead of being recomputed."""
    env = orc.make_env(cfg["env_id"], lq_config=cfg.get("lq_config", "s4a2"),
                       pre_horizon=cfg.get("pre_horizon", 10), surr_veh_num=cfg.get("surr_veh_num"),
                       reward_scale=extra.get("reward_scale"), reward_shift=extra.get("reward_shift"),
                       obs_scale=extra.get("obs_scale"), obs_shift=extra.get("obs_shift"),
                       path_para=extra.get("path_para"), u_para=extra.get("u_para"),
                       repeat_num=extra.get("repeat_num"), sum_reward=extra.get("sum_reward", True),
                       mask_at_done=extra.get("mask_at_done", True))
    if golden is not None and "const/lq_inv_IA" in golden:
        env["lq"]["inv_IA"] = torch.from_numpy(np.array(golden["const/lq_inv_IA"]))
    return env


def data_from_golden(g, prefix="in/"):
    return {k[len(prefix):]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith(prefix)}


def nets_from_golden(g, cfg):
    sd = {k[3:]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith("sd/")}
    act = cfg["act"]
    nets = {}
    pol = orc.net_from_state_dict(sd, "policy.pi", act)
    pol["act_high"], pol["act_low"] = sd["policy.act_high_lim"], sd["policy.act_low_lim"]
    nets["policy"] = pol
    if "v.v.0.weight" in sd:
        nets["v"] = orc.net_from_state_dict(sd, "v.v", act)
        nets["v_target"] = orc.net_from_state_dict(sd, "v_target.v", act, requires_grad=False)
    return nets, sd


def mpg_nets_from_golden(g, cfg):
    """Oracle parameter dicts of an MPG fixture's state_dict (q1, q2, [q1_model, q2_model], targets, policy)."""
    sd = {k[3:]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith("sd/")}
    nets = {}
    for name in ("policy", "policy_target"):
        net = orc.net_from_state_dict(sd, f"{name}.pi", cfg["act"], requires_grad=(name == "policy"))
        net["act_high"], net["act_low"] = sd[f"{name}.act_high_lim"], sd[f"{name}.act_low_lim"]
        nets[name] = net
    for name in ("q1", "q2", "q1_model", "q2_model"):
        if f"{name}.q.0.weight" in sd:
            nets[name] = orc.net_from_state_dict(sd, f"{name}.q", cfg["act"])
            nets[name + "_target"] = orc.net_from_state_dict(sd, f"{name}_target.q", cfg["act"], requires_grad=False)
    return nets, sd


def reference_init_nets(cfg, seed, obs_dim, act_dim):
    """Re-create the reference's random init: torch.manual_seed(seed) then nn.Linear layers in
    the reference's construction order (fhadp.py:42-44; infadp.py:41-45: value first)."""
    torch.manual_seed(seed)

    def linears(sizes):
        return [torch.nn.Linear(sizes[i], sizes[i + 1]) for i in range(len(sizes) - 1)]

    def to_net(layers, **kw):
        return dict(w=[l.weight.detach().clone().requires_grad_(True) for l in layers],
                    b=[l.bias.detach().clone().requires_grad_(True) for l in layers], act=cfg["act"], **kw)

    hid = list(cfg["hidden"])
    lim = dict(act_high=torch.ones(act_dim), act_low=-torch.ones(act_dim))
    nets = {}
    if cfg["alg"] == "FHADP":
        nets["policy"] = to_net(linears([obs_dim + 1] + hid + [act_dim]), **lim)
    else:
        v = linears([obs_dim] + hid + [1])
        p = linears([obs_dim] + hid + [act_dim])
        nets["v"] = to_net(v)
        nets["policy"] = to_net(p, **lim)
        vt = to_net(v)
        g = torch.Generator().manual_seed(seed + 1000)  # make_golden.perturb_targets
        with torch.no_grad():
            for w_, b_ in zip(vt["w"], vt["b"]):
                for p_ in (w_, b_):
                    p_.add_(0.05 * (torch.rand(p_.shape, generator=g) - 0.5))
        for p_ in vt["w"] + vt["b"]:
            p_.requires_grad_(False)
        nets["v_target"] = vt
    return nets


# ---- HIP side: build C-ABI descriptors from the same (oracle-side) constants -------------------
def hip_env_from_oracle(env, policy_net=None):
    from gops_amd import hip_backend as hb
    from gops_amd.env.env_ocp.resources.ref_traj_params import ref_constants
    kind = {"veh_err": hb.ENV_VEH_SURR, "lq": hb.ENV_LQ, "idp": hb.ENV_IDP, "veh": hb.ENV_VEH, "veh_surr": hb.ENV_VEH_SURR,
            "cartpole": hb.ENV_CARTPOLE, "pendulum": hb.ENV_PENDULUM, "veh2": hb.ENV_VEH2DOF, "mob": hb.ENV_MOBILEROBOT}[env["kind"]]
    surr = None
    if env["kind"] in ("veh_surr", "veh_err") or env.get("err_tol") is not None:
        surr = {k: env[k] for k in ("n_surr", "n_constraint", "veh_length", "veh_width", "road_upper", "road_lower", "reward_w")}
        surr["penalty"] = bool(env.get("penalty", False))
        surr["err_tol"] = env.get("err_tol")
    lq = None
    if env["kind"] == "lq":
        c = env["lq"]
        lq = dict(inv_IA=c["inv_IA"], B=c["B"], Q=c["Q"], R=c["R"], dt=c["dt"],
                  reward_scale=c["reward_scale"], reward_shift=c["reward_shift"])
    return hb.make_env(kind, env["obs_dim"], env["act_dim"], act_low=env["act_low"], act_high=env["act_high"],
                       min_action=env["min_action"], max_action=env["max_action"],
                       policy_low=None if policy_net is None else policy_net["act_low"],
                       policy_high=None if policy_net is None else policy_net["act_high"],
                       obs_low=env["obs_low"], obs_high=env["obs_high"], pre_horizon=env.get("P", 0),
                       reward_scale=env["reward_scale"] if env["shaping"] else None,
                       reward_shift=env["reward_shift"] if env["shaping"] else None, lq=lq, surr=surr,
                       obs_scale=env["obs_scale"] if env.get("scale_obs") else None,
                       obs_shift=env["obs_shift"] if env.get("scale_obs") else None,
                       ref_c=ref_constants(env.get("path_para"), env.get("u_para")) if "ref_params" in env else None,
                       repeat_num=env.get("repeat_num"), sum_reward=env.get("sum_reward", True),
                       mask_at_done=env.get("mask_at_done", True),
                       n_constraint=env["n_constraint"] if env["kind"] == "mob" else None)


def hip_mlp_from_net(net, device):
    from gops_amd import hip_backend as hb
    ws = [w.detach().to(device).contiguous() for w in net["w"]]
    bs = [b.detach().to(device).contiguous() for b in net["b"]]
    return hb.make_mlp(ws, bs, net["act"]), ws, bs


def to_device(data, device):
    return {k: v.to(device).contiguous() for k, v in data.items()}


def as_f64(x):
    """Deep copy of an oracle env / net / data structure in float64."""
    if torch.is_tensor(x):
        return x.detach().double() if x.is_floating_point() else x
    if isinstance(x, dict):
        return {k: as_f64(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [as_f64(v) for v in x]
    return x


def fhadp_gradient_f64(env, net, data, horizon, gamma):
    """The oracle's FHADP gradient evaluated in float64: the value both fp32 evaluations (the
    reference's and the HIP path's) approximate.  Used to size the fp32 noise floor of a case."""
    n64 = as_f64(net)
    n64["w"] = [w.requires_grad_(True) for w in n64["w"]]
    n64["b"] = [b.requires_grad_(True) for b in n64["b"]]
    return orc.fhadp_gradient(as_f64(env), n64, as_f64(data), horizon, gamma)


def fp32_noise_floor(env, net, data, horizon, gamma, ref64_flat, trials=8):
    """How far fp32 evaluations of one FHADP gradient scatter around its float64 value: the oracle
    (fp32) is re-run with every weight moved by at most one ulp; returns the largest rel-L2 distance
    to `ref64_flat` (1e-7-ish for well-conditioned cases, up to 5e-4 for the trained LQ H=80 case,
    whose clipped, unstable closed loop amplifies last-bit differences)."""
    gen = torch.Generator().manual_seed(0)
    worst = 0.0
    for _ in range(trials):
        pert = dict(net)
        pert["w"] = [(w.detach() * (1 + (torch.rand(w.shape, generator=gen) - 0.5) * 1.2e-7)).requires_grad_(True)
                     for w in net["w"]]
        pert["b"] = [b.detach().clone().requires_grad_(True) for b in net["b"]]
        grads = orc.fhadp_gradient(env, pert, data, horizon, gamma)["grads"]
        flat = torch.cat([x.reshape(-1) for x in grads]).double()
        worst = max(worst, float((flat - ref64_flat).norm() / ref64_flat.norm()))
    return worst
