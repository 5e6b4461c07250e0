"""GPU: the plane-split kernels (two half planes per operand, 22-bit weights: what every BASELINE shape runs by default) on
256-wide networks TRAINED by the unmodified reference (`tests/golden/make_golden.py trained256`: 300 - 400 `local_update`s of
`gops/algorithm/fhadp.py:87-90` / `infadp.py:101-133`, then one gradient on a held-out batch).  Random-init weights are small
and centred; trained ones have moved (and in the *_sat case saturate the tanh head), so this is where the 16-bit plane
representation has to hold the north_star bar (1e-4 relative L2) - and where it is compared with the exact-fp32 kernels of the
same library on the same fixture.  Every case prints all distances; DESIGN.md section 2 quotes them.

What round 5 measured here (MI355X):
  * the error of the plane-split path sits in the ROLLOUT kernels (forward and sweep: the same perturbed weights for every sample);
    exact-fp32 rollout kernels with the two-half-plane weight-gradient GEMM behind them ("exact_forward" below: what `PrecisionGuard`
    falls back to) are as close to the reference as the all-exact launch - the GEMM's per-sample operand errors average out;
  * a weight error is SYSTEMATIC (the same perturbed network for every sample: it does not average out over the batch).  With the
    round-3 planes (bf16 + f16 residual, 2^-20 |w|) five of the six fixtures sat at 3e-6 .. 3e-5 but the reference-trained pyth_lq
    policy with a saturated tanh head (the REFERENCE's own gradient moves 6.6e-5 under 1-ulp weight moves) at 3.5e-4 - outside
    the bar; with two half planes per operand (2^-22 |w|, csrc/common.h GOPS_SPLIT_F16X2) all six are at the level of the exact-fp32
    kernels: 1.8e-6 .. 6e-6, the saturated one 3.1e-5 (exact forward: 3.2e-5).  `NEEDS_EXACT_FORWARD` is empty; the algorithm
    classes' `PrecisionGuard` (algorithm/base.py) stays as the measured safety net, and catches a half-range overflow;
  * veh3dofconti: the appended reference headings (exemption 1: a 1 ms finite difference in fp32 whose last bit
    depends on the host's libm) reach the GRADIENT of a trained policy - 1.6e-4 (INFADP 256^3 relu) / 4e-5 (FHADP P = 30) whatever
    the arithmetic; with the reference's appended points handed in (`GopsRolloutIn.ref_appended`, here from the oracle's restatement
    on this host) the same launches are at 3e-6 / 1e-5.  The 1e-4 assertions of the veh3dofconti fixtures run in that mode, the
    kernel-evaluated headings are bounded by `VEH_HEADING_BOUND`."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import golden_meta, load_golden, rel_l2
from helpers import data_from_golden, hip_env_from_oracle, nets_from_golden, oracle_env, to_device
from oracle import adp_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-4
VEH_HEADING_BOUND = 5e-4
NEEDS_EXACT_FORWARD = set()   # (round 5, bf16 + f16 planes: the saturated LQ policy, 3.5e-4; two half planes per operand: 3.1e-5)

FHADP_T256 = ["t256_fhadp_idp_h30_gelu", "t256_fhadp_veh_p30_elu", "t256_fhadp_lq_s4a2_elu_sat"]
INFADP_T256 = ["t256_infadp_lq_s4a2_relu", "t256_infadp_lq_s4a2_gelu", "t256_infadp_veh_p10_relu3"]


def bar_of(g, key="meta/ref_fp32_scatter"):
    """1e-4 - unless the REFERENCE's own fp32 gradient of this fixture moves by more than half of that when its weights move by
    one ulp (`make_golden.ref_fp32_scatter`, recorded with the fixture): then twice that scatter.  Of the six fixtures only the
    saturated LQ policy (99.7 % of its first actions beyond 0.99: the gradient flows through the few that are not) is there,
    at 6.6e-5."""
    return max(TOL, 2.0 * float(g[key]))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def _variants():
    from gops_amd import hip_backend as hb
    from gops_amd.algorithm.base import PrecisionGuard
    return {
        "default": 0,                                                  # what bench.py times: plane-split where eligible
        "streamed_split": hb.VF_NO_STATIONARY_SPLIT,                   # the streamed plane-split kernels (B > 4096 takes these)
        "exact_forward": PrecisionGuard.exact_rollout_flags(),         # the guard's fallback: exact forward AND sweep, two-half-plane dW GEMM
        "exact_fp32": hb.VF_STREAMED_FP32 | hb.VF_DW_F32,              # v_mfma_f32_16x16x4_f32 throughout
    }


def _mlp(net, dev, flags):
    from gops_amd import hip_backend as hb
    ws = [w.detach().to(dev).contiguous() for w in net["w"]]
    bs = [b.detach().to(dev).contiguous() for b in net["b"]]
    return hb.make_mlp(ws, bs, net["act"], variant_flags=flags), ws, bs


def _flat(grads):
    return torch.cat([g.reshape(-1).cpu() for g in grads])


def _golden_flat(g, prefix, n):
    return torch.cat([torch.from_numpy(g[f"{prefix}{i}"]).reshape(-1) for i in range(n)])


def _appended_points(cfg, data):
    """The H reference points a veh3dofconti rollout appends, from the oracle's restatement of MultiRefTrajModel on this host
    (same libm as the reference run that recorded the fixture): [B, H, 4]."""
    P, H, dt = cfg["pre_horizon"], cfg["horizon"], 0.1
    t = data["ref_time"].clone()
    pts = []
    for _ in range(H):   # veh_step: nt = ref_time + dt (accumulated in fp32), new point at nt + P dt
        t = t + dt
        pts.append(orc.ref_point(t + P * dt, data["path_num"], data["u_num"]))
    return torch.stack(pts, 1).contiguous()


def _policy_gradient(henv, nets, cfg, ddev, dev, flags, tail):
    from gops_amd import hip_backend as hb
    B = ddev["obs"].shape[0]
    pol, pw, pb = _mlp(nets["policy"], dev, flags)
    vt = _mlp(nets["v_target"], dev, flags)[0] if tail else None
    ro = hb.Rollout(henv, pol, batch=B, horizon=cfg["horizon"], gamma=cfg["gamma"], finite_horizon=not tail, need_grad=True,
                    value=vt, variant_flags=flags)
    variant = hb.lib().gops_rollout_variant(ctypes.byref(ro.desc))
    res = ro.forward(ddev)
    gw, gb = [torch.empty_like(w) for w in pw], [torch.empty_like(b) for b in pb]
    ro.backward(torch.full((B,), -1.0 / B, device=dev), gw, gb)
    torch.cuda.synchronize()
    return variant, -res["v_pi"].double().mean().item(), [t for pair in zip(gw, gb) for t in pair]


def _check_table(name, report, bar, what=""):
    for vname, (dl, err, worst) in report.items():
        split = vname in ("default", "streamed_split")
        assert dl <= TOL, (name, what, vname, dl)
        if split and name in NEEDS_EXACT_FORWARD:
            # the documented case that needs the exact forward: must still be beyond the bar on plane-split kernels, or the
            # exemption (and the guard's reason to exist) has outlived its cause
            assert err > bar, (name, vname, err, "no longer beyond the bar on the plane-split kernels: drop it from NEEDS_EXACT_FORWARD")
        else:
            assert err < bar and worst < bar, (name, what, vname, err, worst, bar)


@pytest.mark.parametrize("name", FHADP_T256)
def test_fhadp_trained_256_wide_vs_reference(name, dev):
    g = load_golden(name)
    meta = golden_meta(g)
    cfg = meta["cfg"]
    env = oracle_env(cfg, meta["extra"], g)
    nets, _ = nets_from_golden(g, cfg)
    data = data_from_golden(g)
    veh = cfg["env_id"] == "pyth_veh3dofconti"
    henv = hip_env_from_oracle(env, nets["policy"])
    modes = {"": data}
    if veh:   # bit-parity mode for the 1e-4 assertions; the kernel's own headings are bounded separately
        modes = {"ref_appended": dict(data, ref_appended=_appended_points(cfg, data)), "kernel_headings": data}
    for mode, d in modes.items():
        ddev = to_device(d, dev)
        report = {}
        for vname, flags in _variants().items():
            variant, loss, grads = _policy_gradient(henv, nets, cfg, ddev, dev, flags, tail=False)
            if vname == "default":
                assert variant == 1, "256-wide FHADP launches of this size take the register-stationary plane-split kernels"
            if vname == "streamed_split" and variant != 4:
                continue   # (veh3dofconti with P = 30: the streamed plane-split kernel does not fit two workgroups' LDS - exact fp32 then)
            if vname in ("exact_forward", "exact_fp32"):
                assert variant not in (1, 4)
            err = rel_l2(_flat(grads), _golden_flat(g, "grad/", len(grads)))
            worst = max(rel_l2(t.cpu(), g[f"grad/{i}"]) for i, t in enumerate(grads))
            report[vname] = (abs(loss - float(g["loss"])) / max(1.0, abs(float(g["loss"]))), err, worst)
        print(f"{name}{' [' + mode + ']' if mode else ''} (policy moved {float(g['meta/policy_moved_rel']):.2f} rel. L2 from init, "
              f"max|w| {float(g['meta/policy_absmax']):.2f}, saturated first actions {float(g['meta/act0_sat_share']):.2f}, reference fp32 "
              f"scatter {float(g['meta/ref_fp32_scatter']):.1e}): "
              + "; ".join(f"{k}: loss {v[0]:.1e} grad {v[1]:.2e} (worst tensor {v[2]:.2e})" for k, v in report.items()))
        _check_table(name, report, VEH_HEADING_BOUND if mode == "kernel_headings" else bar_of(g), mode)


@pytest.mark.parametrize("name", INFADP_T256)
def test_infadp_trained_256_wide_vs_reference(name, dev):
    from gops_amd import hip_backend as hb
    g = load_golden(name)
    meta = golden_meta(g)
    cfg = meta["cfg"]
    env = oracle_env(cfg, meta["extra"], g)
    nets, _ = nets_from_golden(g, cfg)
    data = data_from_golden(g)
    B = data["obs"].shape[0]
    veh = cfg["env_id"] == "pyth_veh3dofconti"
    henv = hip_env_from_oracle(env, nets["policy"])
    modes = {"": data}
    if veh:
        modes = {"ref_appended": dict(data, ref_appended=_appended_points(cfg, data)), "kernel_headings": data}
    for mode, d in modes.items():
        ddev = to_device(d, dev)
        pev_rep, pim_rep = {}, {}
        for vname, flags in _variants().items():
            if vname == "streamed_split" and len(cfg["hidden"]) != 2:
                continue   # three hidden layers: the default IS the streamed plane-split kernel
            vt = _mlp(nets["v_target"], dev, flags)[0]
            v, vw, vb = _mlp(nets["v"], dev, flags)
            pol = _mlp(nets["policy"], dev, flags)[0]
            # PEV: backup from a gradient-free rollout with the tail value, V(o) regression through the value path
            ro = hb.Rollout(henv, pol, batch=B, horizon=cfg["horizon"], gamma=cfg["gamma"], finite_horizon=False, need_grad=False,
                            value=vt, variant_flags=flags)
            backup = ro.forward(ddev)["v_pi"]
            vn = hb.ValueNet(v, B)
            vo = vn.forward(ddev["obs"])
            gw, gb = [torch.empty_like(w) for w in vw], [torch.empty_like(b) for b in vb]
            vn.backward(ddev["obs"], (2.0 / B) * (vo - backup), gw, gb)
            torch.cuda.synchronize()
            pev = [t for pair in zip(gw, gb) for t in pair]
            pev_loss = ((vo - backup).double() ** 2).mean().item()
            del ro, vn
            # PIM: gradient through policy, model and the target value net's input
            variant, pim_loss, pim = _policy_gradient(henv, nets, cfg, ddev, dev, flags, tail=True)
            if vname in ("exact_forward", "exact_fp32"):
                assert variant not in (1, 4)
            else:
                assert variant in (1, 4), "256-wide INFADP launches take plane-split kernels by default"
            pev_rep[vname] = (abs(pev_loss - float(g["pev_loss"])) / max(1.0, abs(float(g["pev_loss"]))),
                              rel_l2(_flat(pev), _golden_flat(g, "pev_grad/", len(pev))),
                              max(rel_l2(t.cpu(), g[f"pev_grad/{i}"]) for i, t in enumerate(pev)))
            pim_rep[vname] = (abs(pim_loss - float(g["pim_loss"])) / max(1.0, abs(float(g["pim_loss"]))),
                              rel_l2(_flat(pim), _golden_flat(g, "pim_grad/", len(pim))),
                              max(rel_l2(t.cpu(), g[f"pim_grad/{i}"]) for i, t in enumerate(pim)))
        print(f"{name}{' [' + mode + ']' if mode else ''} (policy moved {float(g['meta/policy_moved_rel']):.2f} rel. L2 from init, "
              f"max|w| {float(g['meta/policy_absmax']):.2f}, reference fp32 scatter PEV {float(g['meta/ref_fp32_scatter_pev']):.1e} / PIM "
              f"{float(g['meta/ref_fp32_scatter_pim']):.1e}): "
              + "; ".join(f"{k}: PEV grad {pev_rep[k][1]:.2e} (worst {pev_rep[k][2]:.2e}) PIM grad {pim_rep[k][1]:.2e} (worst {pim_rep[k][2]:.2e})"
                          for k in pev_rep))
        heading = mode == "kernel_headings"
        _check_table(name, pev_rep, VEH_HEADING_BOUND if heading else bar_of(g, "meta/ref_fp32_scatter_pev"), mode + " PEV")
        _check_table(name, pim_rep, VEH_HEADING_BOUND if heading else bar_of(g, "meta/ref_fp32_scatter_pim"), mode + " PIM")


# ---- the algorithm classes: PrecisionGuard decides between the plane-split and the exact-fp32 rollout kernels ------------------------
def _load_alg(name, **extra):
    from test_alg_gpu import _kwargs
    from gops_amd.create_pkg.create_alg import create_alg
    g = load_golden(name)
    meta = golden_meta(g)
    cfg = meta["cfg"]
    kw = _kwargs(cfg, meta["extra"], meta["seed"])
    if cfg["alg"] == "FHADP":
        kw["gamma"] = cfg["gamma"]
    kw.update(extra)
    alg = create_alg(**kw)
    sd = {k[3:]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith("sd/")}
    alg.load_state_dict(sd)      # the reference's checkpoint layout loads unchanged
    alg.networks.cuda()
    if cfg["alg"] == "INFADP":
        alg.gamma, alg.forward_step = cfg["gamma"], cfg["horizon"]
    if "const/lq_inv_IA" in g:
        # (I - A dt)^-1 comes out of LAPACK in fp32 with host-dependent last bits (lq_base.py:55-57); the fixture carries the value
        # its outputs were recorded with, and one ulp of THIS constant - applied at every step of every trajectory - is worth
        # ~1e-4 of gradient on these trained policies (measured: class 8e-5 / 1.2e-4 with this host's pinv, 2e-5 with the recorded one)
        dyn = alg.envmodel.unwrapped.dynamics
        dyn.inv_IA = torch.from_numpy(np.array(g["const/lq_inv_IA"])).to(dyn.inv_IA.device)
    return alg, g, cfg


@pytest.mark.parametrize("name", FHADP_T256)
def test_precision_guard_fhadp(name, dev):
    """The FHADP class on the trained fixtures: the first gradient is checked (a loaded checkpoint may already need the exact
    forward).  With two half planes per operand every fixture stays on the plane-split kernels, its measured distance to the
    exact-fp32 rollout kernels far below the threshold; with the threshold set below that distance the guard trips, stays tripped, and
    the class returns the exact kernels' gradient."""
    alg, g, cfg = _load_alg(name)
    data = to_device(data_from_golden(g), dev)
    guard = alg.precision_guard
    _, info = alg.get_remote_update_info(data, 0)
    torch.cuda.synchronize()
    assert guard.checks == 1 and guard.last_distance is not None
    grads = info["grad"]
    err = rel_l2(_flat(grads), _golden_flat(g, "grad/", len(grads)))
    print(f"{name}: guard distance {guard.last_distance:.2e} (threshold {guard.threshold:.0e}) -> exact forward: {guard.exact}; "
          f"gradient of the class vs the reference {err:.2e}")
    assert not guard.exact and guard.last_distance < guard.threshold, (name, guard.last_distance)
    veh = cfg["env_id"] == "pyth_veh3dofconti"   # (the class evaluates the appended headings itself: exemption 1)
    assert err < (VEH_HEADING_BOUND if veh else bar_of(g)), (name, err)
    for it in range(1, 4):   # an unexceptional network is checked again only after `interval` gradients
        alg.get_remote_update_info(data, it)
    assert guard.checks == 1 and not guard.exact
    assert alg._rollout_for(data["obs"].shape[0], dev).desc.variant_flags == guard.flags()
    # the same network under a threshold below its measured distance: trips at the first gradient, and stays on the exact forward
    alg2, _, _ = _load_alg(name, precision_threshold=guard.last_distance * 0.5)
    with pytest.warns(UserWarning, match="exact-fp32 rollout kernels"):
        _, info2 = alg2.get_remote_update_info(data, 0)
    torch.cuda.synchronize()
    g2 = alg2.precision_guard
    assert g2.exact and g2.checks == 1
    err2 = rel_l2(_flat(info2["grad"]), _golden_flat(g, "grad/", len(info2["grad"])))
    assert err2 < (VEH_HEADING_BOUND if veh else bar_of(g)), (name, err2)
    for it in range(1, 4):
        alg2.get_remote_update_info(data, it)
    assert g2.checks == 1 and g2.exact
    from gops_amd.algorithm.base import PrecisionGuard
    flags = alg2._rollout_for(data["obs"].shape[0], dev).desc.variant_flags
    assert flags == g2.flags() and flags & PrecisionGuard.exact_rollout_flags() == PrecisionGuard.exact_rollout_flags()


def test_half_range_overflow_is_loud_and_the_guard_catches_it(dev):
    """The plane-split forward carries activations as two half planes scaled by 2^-4: beyond |a| = 1.05e6 the conversion overflows
    and the rollout returns NON-FINITE values (never silently wrong ones).  A policy whose first layer is scaled so that H_1
    reaches ~1e8: the raw plane-split launch is non-finite; the FHADP class measures a non-finite distance at its first gradient,
    moves to the exact-fp32 rollout kernels and returns the reference gradient."""
    from gops_amd import hip_backend as hb
    from gops_amd.utils.synthetic import act_dim_of, make_batch, obs_dim_of
    from helpers import reference_init_nets
    cfg = dict(alg="FHADP", env_id="pyth_lq", lq_config="s4a2", batch=64, horizon=3, hidden=(256, 256), act="relu", gamma=0.99)
    data = make_batch(cfg, 21)
    nets = reference_init_nets(cfg, 21, obs_dim_of(cfg), act_dim_of(cfg))
    with torch.no_grad():
        nets["policy"]["w"][0].mul_(2.0e8)
        nets["policy"]["b"][0].mul_(2.0e8)
        nets["policy"]["w"][1].mul_(1.0e-8)
    env = orc.make_env("pyth_lq", lq_config="s4a2")
    ref = orc.fhadp_gradient(env, nets["policy"], data, cfg["horizon"], cfg["gamma"])
    ddev = to_device(data, dev)
    henv = hip_env_from_oracle(env, nets["policy"])
    variant, loss, grads = _policy_gradient(henv, nets, cfg, ddev, dev, 0, tail=False)
    assert variant == 1
    assert not torch.isfinite(_flat(grads)).all() or not np.isfinite(loss), "beyond the half range the plane-split launch must not look sane"
    # the class: same weights through the state_dict layout of the reference
    from test_alg_gpu import _kwargs
    from gops_amd.create_pkg.create_alg import create_alg
    alg = create_alg(**_kwargs(cfg, {}, 21), gamma=cfg["gamma"])
    alg.networks.cuda()
    with torch.no_grad():
        for layer, w, b in zip(alg.networks.policy.linear_layers(), nets["policy"]["w"], nets["policy"]["b"]):
            layer.weight.copy_(w.detach())
            layer.bias.copy_(b.detach())
    alg.envmodel.unwrapped.dynamics.inv_IA = env["lq"]["inv_IA"].to(dev)
    with pytest.warns(UserWarning, match="exact-fp32 rollout kernels"):
        _, info = alg.get_remote_update_info(ddev, 0)
    torch.cuda.synchronize()
    assert alg.precision_guard.exact
    got, want = _flat(info["grad"]), torch.cat([t.reshape(-1) for t in ref["grads"]])
    assert torch.isfinite(got).all()
    assert rel_l2(got, want) < TOL, rel_l2(got, want)


class _no_context:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def test_precision_guard_interval_and_switch_off(dev):
    alg, g, cfg = _load_alg("t256_fhadp_idp_h30_gelu", precision_check_interval=3)
    data = to_device(data_from_golden(g), dev)
    for it in range(7):
        alg.get_remote_update_info(data, it)
    assert alg.precision_guard.checks == 3     # gradients 1, 3 and 6
    assert not alg.precision_guard.exact
    alg2, g2, _ = _load_alg("t256_fhadp_lq_s4a2_elu_sat", precision_check_interval=0, precision_threshold=1e-12)
    data2 = to_device(data_from_golden(g2), dev)
    alg2.get_remote_update_info(data2, 0)
    assert alg2.precision_guard.checks == 0 and not alg2.precision_guard.exact   # switched off: the caller's responsibility


@pytest.mark.parametrize("name", ["t256_infadp_lq_s4a2_relu", "t256_infadp_lq_s4a2_gelu"])
def test_precision_guard_infadp(name, dev):
    """INFADP keeps one guard per trained network; on these fixtures both stay on the plane-split kernels."""
    alg, g, cfg = _load_alg(name)
    data = to_device(data_from_golden(g), dev)
    _, info_v = alg.get_remote_update_info(data, 0)     # PEV
    pev = rel_l2(_flat(info_v["v"]), _golden_flat(g, "pev_grad/", len(info_v["v"])))
    _, info_p = alg.get_remote_update_info(data, 1)     # PIM
    pim = rel_l2(_flat(info_p["policy"]), _golden_flat(g, "pim_grad/", len(info_p["policy"])))
    torch.cuda.synchronize()
    gv, gp = alg.precision_guard["v"], alg.precision_guard["policy"]
    print(f"{name}: guard distances PEV {gv.last_distance:.2e} PIM {gp.last_distance:.2e}; class vs reference PEV {pev:.2e} PIM {pim:.2e}")
    assert gv.checks == 1 and gp.checks == 1 and not gv.exact and not gp.exact
    assert pev < bar_of(g, "meta/ref_fp32_scatter_pev") and pim < bar_of(g, "meta/ref_fp32_scatter_pim")


def _overflowing_fhadp(dev, **more):
    """An FHADP object on 256-wide nets whose first layer drives H_1 to ~1e8 (beyond the half range of the plane-split forward),
    its device batch, and the oracle's reference gradient."""
    from gops_amd.utils.synthetic import act_dim_of, make_batch, obs_dim_of
    from helpers import reference_init_nets
    from test_alg_gpu import _kwargs
    from gops_amd.create_pkg.create_alg import create_alg
    cfg = dict(alg="FHADP", env_id="pyth_lq", lq_config="s4a2", batch=64, horizon=3, hidden=(256, 256), act="relu", gamma=0.99)
    data = make_batch(cfg, 21)
    nets = reference_init_nets(cfg, 21, obs_dim_of(cfg), act_dim_of(cfg))
    alg = create_alg(**dict(_kwargs(cfg, {}, 21), gamma=cfg["gamma"], **more))
    alg.networks.cuda()
    with torch.no_grad():
        for layer, w, b in zip(alg.networks.policy.linear_layers(), nets["policy"]["w"], nets["policy"]["b"]):
            layer.weight.copy_(w.detach())
            layer.bias.copy_(b.detach())
    return alg, to_device(data, dev)


def _blow_up(alg):
    with torch.no_grad():
        l0, l1 = alg.networks.policy.linear_layers()[:2]
        l0.weight.mul_(2.0e8)
        l0.bias.mul_(2.0e8)
        l1.weight.mul_(1.0e-8)


def test_overflow_between_guard_checks_leaves_the_weights_intact_and_forces_a_check(dev):
    """The guard checks every `interval` gradients; an overflow in between must not reach the weights.  Healthy first gradient
    (check passes), then the first layer is blown up: the plane-split kernels answer with NaN gradients, the optimizer kernels -
    here the fused tail of the backward's last launch, then the stand-alone Adam of `remote_update` - take no step for them
    (weights and moments bit-identical, `skipped_nonfinite` counts), and reading the logged loss makes the NEXT gradient a
    checked one, which moves the network to the exact-fp32 rollout kernels."""
    import warnings
    alg, data = _overflowing_fhadp(dev, precision_check_interval=1000)
    opt = alg.networks.policy_optimizer
    tb = alg.local_update(data, 0)
    assert np.isfinite(float(tb["Loss/Actor loss-RL iter"])) and alg.precision_guard.checks == 1 and not alg.precision_guard.exact
    assert opt.skipped_nonfinite() == 0
    _blow_up(alg)
    before = [p.detach().clone() for p in alg.networks.policy.parameters()]
    moments = [opt.state[p]["exp_avg"].clone() for p in alg.networks.policy.parameters()]
    tb = alg.local_update(data, 1)                                   # fused tail (gops_rollout_backward_update)
    torch.cuda.synchronize()
    n_params = sum(p.numel() for p in before)
    assert opt.skipped_nonfinite() == n_params
    assert all(torch.equal(a, b) for a, b in zip(alg.networks.policy.parameters(), before))
    assert all(torch.equal(opt.state[p]["exp_avg"], m) for p, m in zip(alg.networks.policy.parameters(), moments))
    lost = tb["Loss/Actor loss-RL iter"]                              # (tb is the algorithm's own dict: keep the entry of THIS update)
    tb2, info = alg.get_remote_update_info(data, 2)                   # the data-parallel path: stand-alone gops_adam_step
    alg.remote_update(info)
    torch.cuda.synchronize()
    assert opt.skipped_nonfinite() == 2 * n_params
    assert all(torch.equal(a, b) for a, b in zip(alg.networks.policy.parameters(), before))
    assert not alg.precision_guard.exact and alg.precision_guard.checks == 1
    assert not np.isfinite(float(lost))                               # reading the lazily logged loss arms the guard ...
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        tb = alg.local_update(data, 3)                               # ... and this gradient is checked
    assert alg.precision_guard.exact and alg.precision_guard.checks == 2
    assert any("exact-fp32 rollout kernels" in str(x.message) for x in w)
    assert np.isfinite(float(tb["Loss/Actor loss-RL iter"]))
    assert any((a - b).abs().max() > 0 for a, b in zip(alg.networks.policy.parameters(), before))   # the exact kernels step again
    assert all(torch.isfinite(p).all() for p in alg.networks.policy.parameters())


def test_a_tripped_guard_drops_the_captured_update_graph(dev):
    """B x H = 192: the update replays as a HIP graph after two eager calls.  The graph bakes the kernel variant in, so a guard
    that trips later must lead to a re-capture on the exact kernels - replaying the plane-split graph would keep producing NaN."""
    import warnings
    alg, data = _overflowing_fhadp(dev, precision_check_interval=6)
    for it in range(4):                                              # check at gradient 1; captured at the third update
        alg.local_update(data, it)
    assert alg._update_graph.graph is not None and not alg.precision_guard.exact
    sig_flags = alg._variant_flags()
    _blow_up(alg)
    before = [p.detach().clone() for p in alg.networks.policy.parameters()]
    alg.local_update(data, 4)                                        # gradient 5: replayed plane-split graph, NaN, no step taken
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(alg.networks.policy.parameters(), before))
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        tb = alg.local_update(data, 5)                               # gradient 6: scheduled check trips
    assert alg.precision_guard.exact and alg._variant_flags() != sig_flags
    assert np.isfinite(float(tb["Loss/Actor loss-RL iter"]))
    for it in range(6, 10):                                          # eager, eager, re-captured, replayed - all on the exact kernels
        tb = alg.local_update(data, it)
        assert np.isfinite(float(tb["Loss/Actor loss-RL iter"])), it
    assert alg._update_graph.graph is not None
    assert all(torch.isfinite(p).all() for p in alg.networks.policy.parameters())
