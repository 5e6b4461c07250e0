"""ctypes binding of libgops_hip.so (C ABI: include/gops_hip.h).

PyTorch is used only for device memory and streams: every call passes raw device pointers of
caller-owned tensors plus the current HIP stream.  There is NO CPU or eager-PyTorch fallback:
if the shared library is missing, or a shape is unsupported, the call raises.
"""
import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GOPS_HIP_LIB: another build of the same library (the phase-counter build `make -C gops_amd/csrc dbg`)
LIB_PATH = os.environ.get("GOPS_HIP_LIB") or os.path.join(_HERE, "libgops_hip.so")

MAX_LAYERS, MAX_ACT, MAX_LQ, TILE = 5, 4, 6, 16
ENV_NONE, ENV_LQ, ENV_IDP, ENV_VEH, ENV_VEH_SURR, ENV_CARTPOLE, ENV_PENDULUM, ENV_VEH2DOF, ENV_MOBILEROBOT = 0, 1, 2, 3, 4, 5, 6, 7, 8
MAX_CLIP_OBS = 16
# std of the obstacle robot's per-step (v, w) noise draws (pyth_mobilerobot_model.py:141-147, std_type["obs"])
MOBILEROBOT_NOISE_STD = (0.03, 0.02)
MAX_SURR = 4
MAX_REPEAT = 8   # GOPS_MAX_REPEAT
ACT_IDS = {"linear": 0, "relu": 1, "elu": 2, "gelu": 3, "selu": 4, "sigmoid": 5, "tanh": 6}
DTYPE_IDS = {"fp32": 0, "f32": 0, "float32": 0, "fp16": 1, "f16": 1, "float16": 1, "half": 1}


def dtype_id(name) -> int:
    """GOPS_DTYPE_* of an `mlp_dtype` setting ("fp32" default; "fp16" = half-precision MFMA contractions)."""
    if name is None:
        return 0
    try:
        return DTYPE_IDS[str(name).lower()]
    except KeyError:
        raise RuntimeError(f"unknown mlp_dtype {name!r}: expected 'fp32' or 'fp16'") from None

_fp = C.POINTER(C.c_float)


class GopsMlp(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("sizes", C.c_int32 * (MAX_LAYERS + 1)),
                ("hidden_act", C.c_int32), ("dtype", C.c_int32), ("variant_flags", C.c_uint32),
                ("weight", C.c_void_p * MAX_LAYERS), ("bias", C.c_void_p * MAX_LAYERS)]


class GopsMlpGrad(C.Structure):
    _fields_ = [("weight", C.c_void_p * MAX_LAYERS), ("bias", C.c_void_p * MAX_LAYERS)]


class GopsEnv(C.Structure):
    _fields_ = [("kind", C.c_int32), ("obs_dim", C.c_int32), ("act_dim", C.c_int32),
                ("pre_horizon", C.c_int32),
                ("min_action", C.c_float * MAX_ACT), ("max_action", C.c_float * MAX_ACT),
                ("act_low", C.c_float * MAX_ACT), ("act_high", C.c_float * MAX_ACT),
                ("policy_low", C.c_float * MAX_ACT), ("policy_high", C.c_float * MAX_ACT),
                ("clip_obs", C.c_int32), ("obs_low", C.c_float * MAX_CLIP_OBS), ("obs_high", C.c_float * MAX_CLIP_OBS),
                ("shaping", C.c_int32), ("reward_scale", C.c_float), ("reward_shift", C.c_float),
                ("lq_inv_IA", C.c_float * (MAX_LQ * MAX_LQ)), ("lq_B", C.c_float * (MAX_LQ * MAX_ACT)),
                ("lq_Q", C.c_float * MAX_LQ), ("lq_R", C.c_float * MAX_ACT),
                ("lq_dt", C.c_float), ("lq_reward_scale", C.c_float), ("lq_reward_shift", C.c_float),
                ("no_mask_at_done", C.c_int32), ("n_surr", C.c_int32), ("n_constraint", C.c_int32), ("surr_penalty", C.c_int32),
                ("veh_length", C.c_float), ("veh_width", C.c_float),
                ("road_upper", C.c_float), ("road_lower", C.c_float), ("reward_w", C.c_float * 8),
                ("data_env", C.c_int32), ("scale_obs", C.c_int32), ("obs_scale", C.c_float * 8), ("obs_shift", C.c_float * 8),
                ("cstr_err", C.c_int32), ("err_tol", C.c_float * 2),
                ("ref_custom", C.c_int32), ("ref_c", C.c_float * 24),
                ("repeat_num", C.c_int32), ("repeat_last_reward", C.c_int32)]


class GopsRolloutDesc(C.Structure):
    _fields_ = [("batch", C.c_int32), ("horizon", C.c_int32), ("finite_horizon", C.c_int32),
                ("need_grad", C.c_int32), ("tail_value", C.c_int32), ("open_loop", C.c_int32),
                ("dtype", C.c_int32), ("tail_unmasked", C.c_int32), ("variant_flags", C.c_uint32), ("l2_warmup", C.c_int32),
                ("dw_workgroups", C.c_int32), ("reserved0", C.c_int32), ("gamma", C.c_double), ("env", GopsEnv), ("policy", GopsMlp),
                ("value", GopsMlp)]


# GopsRolloutDesc.variant_flags / GopsMlp.variant_flags (include/gops_hip.h, ABI v10): kernel-variant selection is part of the
# description - no process-global state.  DEFAULT_VARIANT_FLAGS is what `Rollout` / `make_mlp` use when the caller passes none
# (tests patch it to steer the algorithm classes).
VF_NO_STATIONARY_SPLIT, VF_NO_STREAMED_SPLIT_FWD, VF_NO_STREAMED_SPLIT_BWD, VF_NO_STREAMED_SPLIT_VALUE = 0x1, 0x2, 0x4, 0x8
VF_STREAMED_FP32, VF_STREAM_LAYER0, VF_STATIONARY_ANY_BATCH, VF_NO_SPLIT_STREAM0, VF_SPLIT_TAIL_MULTI = 0x10, 0x20, 0x40, 0x80, 0x100
VF_NO_HALF_TILE64, VF_NO_NARROW_LDS, VF_NO_NARROW_N64 = 0x200, 0x400, 0x800
VF_DW_EXACT, VF_DW_F32, VF_DW_NO_GUARD, VF_DW_NO_SKINNY, VF_DW_NO_SPEC, VF_DW_DIRECT = 0x10000, 0x20000, 0x40000, 0x80000, 0x100000, 0x200000
VF_NO_FUSED_DWOUT, VF_BWD_UPLOAD = 0x400000, 0x800000
VF_BWD_PHASE_A, VF_BWD_PHASE_B, VF_NO_FUSED_DW0 = 0x1000000, 0x2000000, 0x4000000   # a backward in two halves (overlapped gradient all-reduce, trainer/grad_sync.py)
DEFAULT_VARIANT_FLAGS = 0


class GopsRolloutIn(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("obs", "done", "state", "ref_points", "path_num", "u_num",
                                          "ref_time", "head_pre", "surr_state", "grad_constraint", "grad_constraint_prod",
                                          "ref_appended", "noise", "grad_constraint_step")]


class GopsRolloutOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("v_pi", "rewards", "final_obs", "final_done", "final_state", "constraint_sums",
                                          "constraint_prods", "constraints")]


ADAM_MAX = 16


class GopsRolloutAdjoint(C.Structure):
    _fields_ = [("grad_final_obs", C.c_void_p), ("grad_obs", C.c_void_p), ("first_step_only", C.c_int32),
                ("reserved", C.c_int32)]


class GopsAdamTensors(C.Structure):
    _fields_ = [("n", C.c_int32), ("reserved", C.c_int32), ("numel", C.c_int64 * ADAM_MAX),
                ("param", C.c_void_p * ADAM_MAX), ("grad", C.c_void_p * ADAM_MAX),
                ("exp_avg", C.c_void_p * ADAM_MAX), ("exp_avg_sq", C.c_void_p * ADAM_MAX)]


class GopsUpdateTail(C.Structure):   # ABI v12: gops_rollout_backward_update
    _fields_ = [("adam", C.POINTER(GopsAdamTensors)), ("adam_state", C.c_void_p), ("beta1", C.c_double), ("beta2", C.c_double),
                ("eps", C.c_double), ("mean_x", C.c_void_p), ("mean_n", C.c_int32), ("reserved", C.c_int32),
                ("mean_scale", C.c_double), ("mean_stats", C.c_void_p), ("polyak", C.POINTER(GopsAdamTensors)), ("polyak_tau", C.c_double)]


class GopsStepIO(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("obs", "action", "done", "state", "ref_points", "path_num",
                                          "u_num", "ref_time", "next_obs", "reward", "next_done",
                                          "next_state", "next_ref_points", "next_ref_time",
                                          "surr_state", "next_surr_state", "constraint", "ref_appended", "noise")]


_lib = None


def lib() -> C.CDLL:
    """Load libgops_hip.so; fails loudly (no fallback) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C gops_amd/csrc`). The HIP rollout has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        l.gops_hip_version.restype = C.c_int
        l.gops_rollout_workspace_bytes.restype = C.c_size_t
        l.gops_rollout_workspace_bytes.argtypes = [C.POINTER(GopsRolloutDesc)]
        l.gops_rollout_forward.restype = C.c_int
        l.gops_rollout_forward.argtypes = [C.POINTER(GopsRolloutDesc), C.POINTER(GopsRolloutIn),
                                           C.POINTER(GopsRolloutOut), C.c_void_p, C.c_size_t, C.c_void_p]
        l.gops_rollout_backward_open_loop.restype = C.c_int
        l.gops_rollout_backward_open_loop.argtypes = [C.POINTER(GopsRolloutDesc), C.POINTER(GopsRolloutIn), C.c_void_p,
                                                      C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        l.gops_rollout_backward.restype = C.c_int
        l.gops_rollout_backward.argtypes = [C.POINTER(GopsRolloutDesc), C.POINTER(GopsRolloutIn), C.c_void_p,
                                            C.POINTER(GopsMlpGrad), C.c_void_p, C.c_size_t, C.c_void_p]
        l.gops_rollout_backward_update.restype = C.c_int
        l.gops_rollout_backward_update.argtypes = [C.POINTER(GopsRolloutDesc), C.POINTER(GopsRolloutIn), C.c_void_p,
                                                   C.POINTER(GopsMlpGrad), C.POINTER(GopsUpdateTail), C.c_void_p, C.c_size_t, C.c_void_p]
        l.gops_value_backward_update.restype = C.c_int
        l.gops_value_backward_update.argtypes = [C.POINTER(GopsMlp), C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(GopsMlpGrad),
                                                 C.POINTER(GopsUpdateTail), C.c_void_p, C.c_size_t, C.c_void_p]
        l.gops_rollout_backward_adj.restype = C.c_int
        l.gops_rollout_backward_adj.argtypes = [C.POINTER(GopsRolloutDesc), C.POINTER(GopsRolloutIn), C.c_void_p,
                                                C.POINTER(GopsMlpGrad), C.POINTER(GopsRolloutAdjoint), C.c_void_p,
                                                C.c_size_t, C.c_void_p]
        l.gops_env_step.restype = C.c_int
        l.gops_env_step.argtypes = [C.POINTER(GopsEnv), C.c_int32, C.POINTER(GopsStepIO), C.c_void_p]
        l.gops_env_constraint.restype = C.c_int
        l.gops_env_constraint.argtypes = [C.POINTER(GopsEnv), C.c_int32, C.POINTER(GopsStepIO), C.c_void_p]
        l.gops_value_workspace_bytes.restype = C.c_size_t
        l.gops_value_workspace_bytes.argtypes = [C.POINTER(GopsMlp), C.c_int32]
        l.gops_value_forward.restype = C.c_int
        l.gops_value_forward.argtypes = [C.POINTER(GopsMlp), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_void_p]
        l.gops_value_backward.restype = C.c_int
        l.gops_value_backward.argtypes = [C.POINTER(GopsMlp), C.c_int32, C.c_void_p, C.c_void_p,
                                          C.POINTER(GopsMlpGrad), C.c_void_p, C.c_size_t, C.c_void_p]
        l.gops_mlp_workspace_bytes.restype = C.c_size_t
        l.gops_mlp_workspace_bytes.argtypes = [C.POINTER(GopsMlp), C.c_int32]
        l.gops_mlp_forward.restype = C.c_int
        l.gops_mlp_forward.argtypes = [C.POINTER(GopsMlp), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        l.gops_mlp_backward.restype = C.c_int
        l.gops_mlp_backward.argtypes = [C.POINTER(GopsMlp), C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(GopsMlpGrad),
                                        C.c_void_p, C.c_size_t, C.c_void_p]
        l.gops_mlp_backward_x.restype = C.c_int
        l.gops_mlp_backward_x.argtypes = [C.POINTER(GopsMlp), C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(GopsMlpGrad),
                                          C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        l.gops_adam_step.restype = C.c_int
        l.gops_adam_step.argtypes = [C.POINTER(GopsAdamTensors), C.c_void_p, C.c_double, C.c_double, C.c_double,
                                     C.c_void_p]
        l.gops_polyak_update.restype = C.c_int
        l.gops_polyak_update.argtypes = [C.POINTER(GopsAdamTensors), C.c_double, C.c_void_p]
        l.gops_value_loss.restype = C.c_int
        l.gops_value_loss.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        l.gops_mean_loss.restype = C.c_int
        l.gops_mean_loss.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p]
        l.gops_rollout_backward_open_loop_adj.restype = C.c_int
        l.gops_rollout_backward_open_loop_adj.argtypes = [C.POINTER(GopsRolloutDesc), C.POINTER(GopsRolloutIn), C.c_void_p, C.c_void_p,
                                                          C.POINTER(GopsRolloutAdjoint), C.c_void_p, C.c_size_t, C.c_void_p]
        l.gops_rollout_variant.restype = C.c_int
        l.gops_rollout_variant.argtypes = [C.POINTER(GopsRolloutDesc)]
        l.gops_profile_enable.argtypes = [C.c_int32]
        l.gops_profile_reset.argtypes = []
        l.gops_profile_read.restype = C.c_int
        l.gops_profile_read.argtypes = [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        _lib = l
    return _lib


EXPORTED_SYMBOLS = ("gops_hip_version", "gops_rollout_workspace_bytes", "gops_rollout_forward",
                    "gops_rollout_backward", "gops_rollout_backward_open_loop", "gops_rollout_backward_adj", "gops_env_step", "gops_value_workspace_bytes",
                    "gops_value_forward", "gops_value_backward", "gops_mlp_workspace_bytes", "gops_mlp_forward",
                    "gops_mlp_backward", "gops_mlp_backward_x", "gops_adam_step", "gops_profile_enable",
                    "gops_profile_reset", "gops_profile_read", "gops_rollout_variant", "gops_rollout_backward_open_loop_adj",
                    "gops_env_constraint", "gops_polyak_update", "gops_value_loss", "gops_mean_loss", "gops_rollout_backward_update",
                    "gops_value_backward_update")

_ERR = {-1: "GOPS_ERR_BAD_ARG", -2: "GOPS_ERR_UNSUPPORTED", -3: "GOPS_ERR_WORKSPACE"}


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {_ERR.get(rc, 'hipError_t ' + str(rc))}")


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "need contiguous fp32 device tensor"
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _fill(arr, vals):
    for i, v in enumerate(vals):
        arr[i] = float(v)


def make_mlp(weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], act: str, dtype=None,
             variant_flags: Optional[int] = None) -> GopsMlp:
    """`dtype` ("fp32" / "fp16") and `variant_flags` (VF_*) only matter for `ValueNet` / `Mlp` batches; a `Rollout` takes its
    own for all nets."""
    m = GopsMlp()
    m.dtype = dtype_id(dtype)
    m.variant_flags = DEFAULT_VARIANT_FLAGS if variant_flags is None else variant_flags
    m.n_layers = len(weights)
    if not 2 <= len(weights) <= MAX_LAYERS:
        raise RuntimeError(f"MLP with {len(weights)} Linear layers is outside the HIP path (2..{MAX_LAYERS})")
    m.sizes[0] = weights[0].shape[1]
    for j, (w, b) in enumerate(zip(weights, biases)):
        m.sizes[j + 1] = w.shape[0]
        m.weight[j] = _ptr(w)
        m.bias[j] = _ptr(b)
    m.hidden_act = ACT_IDS[act]
    m._keep = (list(weights), list(biases))   # the struct holds raw pointers: keep the tensors alive
    return m


def make_mlp_grad(gw: Sequence[torch.Tensor], gb: Sequence[torch.Tensor]) -> GopsMlpGrad:
    g = GopsMlpGrad()
    for j, (w, b) in enumerate(zip(gw, gb)):
        g.weight[j] = _ptr(w)
        g.bias[j] = _ptr(b)
    return g


def make_env(kind: int, obs_dim: int, act_dim: int, *, act_low, act_high, min_action=-1.0, max_action=1.0,
             policy_low=None, policy_high=None, obs_low=None, obs_high=None, pre_horizon: int = 0,
             reward_scale: Optional[float] = None, reward_shift: Optional[float] = None,
             lq: Optional[Dict] = None, data_env: bool = False, surr: Optional[Dict] = None,
             obs_scale=None, obs_shift=None, ref_c=None, repeat_num: Optional[int] = None, sum_reward: bool = True,
             mask_at_done: bool = True, n_constraint: Optional[int] = None) -> GopsEnv:
    """Constants of the wrapped env model (create_env_model.py:86-128) as a C struct.  `data_env=True` (for
    `env_step` only): the DATA environment's termination tests / terminal penalty instead of the model's; obs_low /
    obs_high are then the data env's state bounds (pyth_lq) and are NOT applied as a clip."""
    e = GopsEnv()
    e.data_env = int(bool(data_env))
    e.no_mask_at_done = int(not mask_at_done)   # create_env_model(mask_at_done=False): no MaskAtDoneModel in the chain
    if repeat_num is not None:   # ActionRepeatModel (repeat_num = 1 is the identity wrapper)
        e.repeat_num, e.repeat_last_reward = int(repeat_num), int(not sum_reward)
    if ref_c is not None:   # custom path_para / u_para of the reference trajectories (resources/ref_traj_params.py)
        assert len(ref_c) == 24
        e.ref_custom = 1
        _fill(e.ref_c, [float(v) for v in ref_c])
    if obs_scale is not None or obs_shift is not None:   # ScaleObservationModel: obs seen = (obs + shift) * scale
        if obs_dim > 8:
            raise RuntimeError("obs_scale / obs_shift: observation dimension > 8 is not supported by the HIP env models")

        def bo(v, default):
            v = torch.as_tensor(default if v is None else v, dtype=torch.float32).reshape(-1)
            if v.numel() not in (1, obs_dim):
                raise RuntimeError(f"obs_scale / obs_shift must be scalars or have {obs_dim} entries")
            return (v.expand(obs_dim) if v.numel() == 1 else v).tolist() + [0.0] * (8 - obs_dim)
        e.scale_obs = 1
        _fill(e.obs_scale, bo(obs_scale, 1.0)); _fill(e.obs_shift, bo(obs_shift, 0.0))
        for i in range(obs_dim, 8):
            e.obs_scale[i] = 1.0
    if surr is not None:   # ENV_VEH_SURR: surrounding vehicles, constraint geometry, reward weights
        e.n_surr, e.n_constraint = int(surr["n_surr"]), int(surr["n_constraint"])
        e.veh_length, e.veh_width = float(surr["veh_length"]), float(surr["veh_width"])
        e.road_upper, e.road_lower = float(surr.get("road_upper", 0.0)), float(surr.get("road_lower", 0.0))
        e.surr_penalty = int(bool(surr.get("penalty", False)))
        if surr.get("err_tol") is not None:   # pyth_veh3dofconti_errcstr: constraints on the tracking errors of the observation
            e.cstr_err = 1
            _fill(e.err_tol, [float(v) for v in surr["err_tol"]])
        _fill(e.reward_w, list(surr["reward_w"]) + [0.0] * (8 - len(surr["reward_w"])))
    e.kind, e.obs_dim, e.act_dim, e.pre_horizon = kind, obs_dim, act_dim, pre_horizon
    if n_constraint is not None:
        e.n_constraint = int(n_constraint)
    A = act_dim

    def bc(v):
        v = torch.as_tensor(v, dtype=torch.float32).reshape(-1)
        return (v.expand(A) if v.numel() == 1 else v).tolist()

    _fill(e.min_action, bc(min_action)); _fill(e.max_action, bc(max_action))
    _fill(e.act_low, bc(act_low)); _fill(e.act_high, bc(act_high))
    _fill(e.policy_low, bc(-1.0 if policy_low is None else policy_low))
    _fill(e.policy_high, bc(1.0 if policy_high is None else policy_high))
    finite = False
    if obs_low is not None:
        lo = torch.as_tensor(obs_low, dtype=torch.float32).reshape(-1)
        hi = torch.as_tensor(obs_high, dtype=torch.float32).reshape(-1)
        finite = bool(torch.isfinite(lo).any() or torch.isfinite(hi).any())
        if finite:
            if obs_dim > (MAX_CLIP_OBS if kind == ENV_MOBILEROBOT else 8):
                raise RuntimeError("finite observation bounds are only supported for obs_dim <= 8 (pyth_lq, gym models) "
                                   "and for pyth_mobilerobot")
            _fill(e.obs_low, lo.tolist()); _fill(e.obs_high, hi.tolist())
    e.clip_obs = 1 if finite else 0
    e.shaping = 1 if (reward_scale is not None or reward_shift is not None) else 0
    e.reward_scale = 1.0 if reward_scale is None else float(reward_scale)
    e.reward_shift = 0.0 if reward_shift is None else float(reward_shift)
    if lq is not None:
        n, m = obs_dim, act_dim
        _fill(e.lq_inv_IA, torch.as_tensor(lq["inv_IA"], dtype=torch.float32).reshape(n * n).tolist())
        _fill(e.lq_B, torch.as_tensor(lq["B"], dtype=torch.float32).reshape(n * m).tolist())
        _fill(e.lq_Q, torch.as_tensor(lq["Q"], dtype=torch.float32).tolist())
        _fill(e.lq_R, torch.as_tensor(lq["R"], dtype=torch.float32).tolist())
        e.lq_dt = float(lq["dt"])
        e.lq_reward_scale = float(lq.get("reward_scale", 1.0))
        e.lq_reward_shift = float(lq.get("reward_shift", 0.0))
    return e


def has_constraints(env: GopsEnv) -> bool:
    """Models whose rollout returns constraint sums / products (and whose env step fills info["constraint"])."""
    return env.kind in (ENV_VEH_SURR, ENV_MOBILEROBOT) or (env.kind == ENV_VEH2DOF and env.cstr_err != 0)


def mobilerobot_noise(shape, device) -> torch.Tensor:
    """The obstacle robot's noise draws of `shape` = (..., 2): N(0, 0.03) for v, N(0, 0.02) for w - what
    np.random.normal hands Robot.f_xu(.., "obs") every step (pyth_mobilerobot_model.py:141-167), drawn on the device."""
    n = torch.randn(*shape, dtype=torch.float32, device=device)
    n[..., 0] *= MOBILEROBOT_NOISE_STD[0]   # (scalar kernel arguments: no host-to-device copy, so a HIP-graph capture of the
    n[..., 1] *= MOBILEROBOT_NOISE_STD[1]   #  caller stays legal and every replay draws fresh numbers from the captured generator)
    return n


class Rollout:
    """One configured horizon rollout (forward + backward) bound to caller-owned tensors.

    The workspace (activation stash, packed weights, reference table, split-K partials) is a
    single torch uint8 tensor sized by `gops_rollout_workspace_bytes` and reused across calls.
    """

    def __init__(self, env: GopsEnv, policy: Optional[GopsMlp], *, batch: int, horizon: int, gamma: float,
                 finite_horizon: bool, need_grad: bool = True, value: Optional[GopsMlp] = None,
                 device: Optional[torch.device] = None, dtype=None, raw_actions: bool = False,
                 tail_unmasked: bool = False, variant_flags: Optional[int] = None, l2_warmup: int = 0, dw_workgroups: int = 0):
        """`policy=None` selects the open-loop mode: `forward(data, head_pre=...)` takes the pre-tanh
        policy-head outputs of all steps [B, H, act_dim] and `backward_open_loop` returns their gradient.
        `dtype`: "fp32" (default: fp32 results at the 1e-4 bar, plane-split MFMAs where they apply) or "fp16" (half-precision
        MFMA contractions and stash).  `variant_flags`: VF_* bits (GopsRolloutDesc.variant_flags), None = DEFAULT_VARIANT_FLAGS."""
        self.desc = GopsRolloutDesc()
        d = self.desc
        d.dtype = dtype_id(dtype)
        d.variant_flags = DEFAULT_VARIANT_FLAGS if variant_flags is None else variant_flags
        d.l2_warmup, d.dw_workgroups = int(l2_warmup), int(dw_workgroups)
        d.batch, d.horizon, d.finite_horizon = batch, horizon, int(finite_horizon)
        d.need_grad, d.tail_value, d.gamma = int(need_grad), int(value is not None), float(gamma)
        d.tail_unmasked = int(bool(tail_unmasked))   # SPIL's evaluation target: the terminal value is not masked at done
        d.env = env
        # raw_actions (open loop only): `head_pre` holds the model's actions themselves (GopsRolloutDesc.open_loop = 2)
        d.open_loop = (2 if raw_actions else 1) if policy is None else 0
        if policy is not None:
            d.policy = policy
        if value is not None:
            d.value = value
        self._mlps = (policy, value)
        nbytes = lib().gops_rollout_workspace_bytes(C.byref(d))
        if nbytes == 0:
            raise RuntimeError("gops_rollout_workspace_bytes: descriptor rejected (unsupported shape for the HIP path)")
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._in = GopsRolloutIn()
        self._keep = None

    def set_policy(self, policy: GopsMlp, value: Optional[GopsMlp] = None):
        """Rebind the networks (their storage moved, or a new GopsMlp was built for them)."""
        if policy is not self._mlps[0]:
            self.desc.policy = policy
        if value is not None and value is not self._mlps[1]:
            self.desc.value = value
        self._mlps = (policy, value if value is not None else self._mlps[1])

    def forward(self, data: Dict[str, torch.Tensor], *, want_rewards=False, want_final=False,
                head_pre: Optional[torch.Tensor] = None, want_constraints=False):
        d = self.desc
        B, H, O = d.batch, d.horizon, d.env.obs_dim
        i = self._in
        if d.open_loop:
            assert head_pre is not None and tuple(head_pre.shape) == (B, H, d.env.act_dim)
            i.head_pre = _ptr(head_pre)
            self._head_pre = head_pre
        i.obs, i.done = _ptr(data["obs"]), _ptr(data.get("done"))
        if d.env.kind in (ENV_VEH, ENV_VEH_SURR, ENV_VEH2DOF):
            for k in ("state", "ref_points", "path_num", "u_num", "ref_time"):
                setattr(i, k, _ptr(data[k]))
        if d.env.kind == ENV_VEH_SURR:
            i.surr_state = _ptr(data.get("surr_state"))   # (None for the errcstr model: no surrounding vehicles)
        # bit-parity mode (GopsRolloutIn.ref_appended): the reference's own appended reference points [B, H, 4]
        i.ref_appended = _ptr(data.get("ref_appended"))
        self._keep = dict(data)
        self._last_phase = None   # (a new forward invalidates a pending backward phase A)
        if d.env.kind == ENV_MOBILEROBOT:   # the obstacle's draws of this rollout [H, B, 2] (a caller may hand in its own)
            noise = data.get("noise")
            if noise is None:
                noise = mobilerobot_noise((H, B, 2), self.device)
            assert tuple(noise.shape) == (H, B, 2) and noise.dtype == torch.float32 and noise.is_contiguous()
            self._keep["noise"] = noise
            i.noise = _ptr(noise)   # the kernels (and a later backward) read these tensors: keep them alive
        out = GopsRolloutOut()
        res = {"v_pi": torch.empty(B, dtype=torch.float32, device=self.device)}
        out.v_pi = _ptr(res["v_pi"])
        if want_rewards:
            res["rewards"] = torch.empty(H, B, dtype=torch.float32, device=self.device)
            out.rewards = _ptr(res["rewards"])
        if want_final:
            res["final_obs"] = torch.empty(B, O, dtype=torch.float32, device=self.device)
            res["final_done"] = torch.empty(B, dtype=torch.float32, device=self.device)
            out.final_obs, out.final_done = _ptr(res["final_obs"]), _ptr(res["final_done"])
            if d.env.kind in (ENV_VEH, ENV_VEH_SURR, ENV_VEH2DOF):
                res["final_state"] = torch.empty(B, 4 if d.env.kind == ENV_VEH2DOF else 6, dtype=torch.float32, device=self.device)
                out.final_state = _ptr(res["final_state"])
        if has_constraints(d.env):   # [4, B]: sum c+^2, sum c+, sum log(-c- + eps), feasible  (discounted, unmasked)
            res["constraint_sums"] = torch.empty(4, B, dtype=torch.float32, device=self.device)
            out.constraint_sums = _ptr(res["constraint_sums"])
            # [2 n_c, B]: prod_t Phi(c_tk) and prod_t [c_tk <= 0] per constraint k (SPIL)
            nc = d.env.n_constraint
            res["constraint_prods"] = torch.empty(2 * nc, B, dtype=torch.float32, device=self.device)
            out.constraint_prods = _ptr(res["constraint_prods"])
            if want_constraints:   # [H, B, n_c]: the unmasked info["constraint"] of every step
                res["constraints"] = torch.empty(H, B, nc, dtype=torch.float32, device=self.device)
                out.constraints = _ptr(res["constraints"])
        check(lib().gops_rollout_forward(C.byref(d), C.byref(i), C.byref(out), self.workspace.data_ptr(),
                                         self.workspace.numel(), _stream()), "gops_rollout_forward")
        return res

    def backward(self, grad_v: torch.Tensor, grad_w: List[torch.Tensor], grad_b: List[torch.Tensor],
                 grad_constraint: Optional[torch.Tensor] = None, grad_constraint_prod: Optional[torch.Tensor] = None,
                 grad_constraint_step: Optional[torch.Tensor] = None, phase: Optional[str] = None,
                 tail: Optional["GopsUpdateTail"] = None):
        """`grad_constraint` (models with constraint outputs): d(loss)/d(constraint_sums rows 0..2), [3, B];
        `grad_constraint_prod`: d(loss)/d(P_k) * P_k for the Phi-products P_k = constraint_prods[k], [n_constraint, B];
        `grad_constraint_step`: d(loss)/d(per-step constraint values), [H, B, n_constraint].
        `phase`: None = the whole backward; "a" = sweep + every gradient except the first hidden layer's, "b" (after "a", same
        arguments) = the first hidden layer's (GOPS_VF_BWD_PHASE_A / _B): lets a data-parallel trainer start the all-reduce of the
        gradients that are ready first while the rest is being formed.
        `tail` (ABI v12, `gops_rollout_backward_update`; with `phase="a"` only a loss-mean tail): the Adam step and / or the loss mean of a single-process
        update, folded into the launch that forms the final gradients (`HipAdam.begin_fused`, `make_update_tail`)."""
        if tail is not None and phase is not None and (phase != "a" or tail.adam or tail.polyak):
            raise ValueError("Rollout.backward: only a loss-mean tail can ride on half a backward, and only on phase 'a'")
        if phase == "b":
            # phase B reuses the inputs (and the delta stash) phase A left in this workspace: it must follow one, directly
            if getattr(self, "_last_phase", None) != "a":
                raise RuntimeError("Rollout.backward(phase='b') must directly follow backward(phase='a') of the same rollout")
            self._last_phase = "b"
            return self._backward_call(grad_v, make_mlp_grad(grad_w, grad_b), VF_BWD_PHASE_B)
        self._last_phase = phase
        g = make_mlp_grad(grad_w, grad_b)
        self._in.grad_constraint = _ptr(grad_constraint)
        self._in.grad_constraint_prod = _ptr(grad_constraint_prod)
        self._in.grad_constraint_step = _ptr(grad_constraint_step)
        self._grad_c = (grad_constraint, grad_constraint_prod, grad_constraint_step)
        self._backward_call(grad_v, g, VF_BWD_PHASE_A if phase == "a" else 0, tail)

    def _backward_call(self, grad_v, g, phase_bits, tail=None):
        flags = self.desc.variant_flags
        self.desc.variant_flags = flags | phase_bits
        try:
            if tail is not None:
                check(lib().gops_rollout_backward_update(C.byref(self.desc), C.byref(self._in), _ptr(grad_v), C.byref(g), C.byref(tail),
                                                         self.workspace.data_ptr(), self.workspace.numel(), _stream()),
                      "gops_rollout_backward_update")
            else:
                check(lib().gops_rollout_backward(C.byref(self.desc), C.byref(self._in), _ptr(grad_v), C.byref(g),
                                                  self.workspace.data_ptr(), self.workspace.numel(), _stream()),
                      "gops_rollout_backward")
        finally:
            self.desc.variant_flags = flags


    def backward_adj(self, grad_v: torch.Tensor, grad_w: Optional[List[torch.Tensor]] = None,
                     grad_b: Optional[List[torch.Tensor]] = None, *, grad_final_obs: Optional[torch.Tensor] = None,
                     want_grad_obs: bool = False, first_step_only: bool = False) -> Optional[torch.Tensor]:
        """`gops_rollout_backward_adj`: the sweep seeded with d(loss)/d(final_obs), optionally returning
        d(loss)/d(obs) of the initial observation; `first_step_only`: parameter gradients through the action of step 0
        only (MPG's model return, mpg.py:340-353).  `grad_w is None`: no parameter gradients."""
        d = self.desc
        self._in.grad_constraint = self._in.grad_constraint_prod = self._in.grad_constraint_step = None   # (no stale seeds of an earlier call)
        adj = GopsRolloutAdjoint()
        adj.grad_final_obs = _ptr(grad_final_obs)
        g_obs = torch.empty(d.batch, d.env.obs_dim, dtype=torch.float32, device=self.device) if want_grad_obs else None
        adj.grad_obs = _ptr(g_obs)
        adj.first_step_only = int(bool(first_step_only))
        g = make_mlp_grad(grad_w, grad_b) if grad_w is not None else None
        check(lib().gops_rollout_backward_adj(C.byref(d), C.byref(self._in), _ptr(grad_v), C.byref(g) if g is not None else None,
                                              C.byref(adj), self.workspace.data_ptr(), self.workspace.numel(), _stream()),
              "gops_rollout_backward_adj")
        return g_obs

    def backward_open_loop_adj(self, grad_v: torch.Tensor, grad_final_obs: Optional[torch.Tensor] = None):
        """`gops_rollout_backward_open_loop_adj`: (d(loss)/d(head_pre) [B, H, act_dim], d(loss)/d(obs) [B, obs_dim]) of the last
        open-loop forward, the sweep seeded with d(loss)/d(final_obs) (models whose observation is the state)."""
        d = self.desc
        self._in.grad_constraint = self._in.grad_constraint_prod = self._in.grad_constraint_step = None
        g = torch.empty(d.batch, d.horizon, d.env.act_dim, dtype=torch.float32, device=self.device)
        g_obs = torch.empty(d.batch, d.env.obs_dim, dtype=torch.float32, device=self.device)
        adj = GopsRolloutAdjoint()
        adj.grad_final_obs, adj.grad_obs, adj.first_step_only = _ptr(grad_final_obs), _ptr(g_obs), 0
        check(lib().gops_rollout_backward_open_loop_adj(C.byref(d), C.byref(self._in), _ptr(grad_v), _ptr(g), C.byref(adj),
                                                        self.workspace.data_ptr(), self.workspace.numel(), _stream()),
              "gops_rollout_backward_open_loop_adj")
        return g, g_obs

    def backward_open_loop(self, grad_v: torch.Tensor, grad_constraint_step: Optional[torch.Tensor] = None) -> torch.Tensor:
        """d(loss)/d(head_pre) [B, H, act_dim] of the last open-loop forward; `grad_constraint_step` [H, B, n_constraint]:
        d(loss)/d(per-step constraint values) (GopsRolloutIn.grad_constraint_step)."""
        d = self.desc
        self._in.grad_constraint = self._in.grad_constraint_prod = None
        self._in.grad_constraint_step = _ptr(grad_constraint_step)
        self._grad_cs = grad_constraint_step
        g = torch.empty(d.batch, d.horizon, d.env.act_dim, dtype=torch.float32, device=self.device)
        check(lib().gops_rollout_backward_open_loop(C.byref(d), C.byref(self._in), _ptr(grad_v), _ptr(g),
                                                    self.workspace.data_ptr(), self.workspace.numel(), _stream()),
              "gops_rollout_backward_open_loop")
        return g


class ValueNet:
    """StateValue batch forward/backward on the HIP path (INFADP PEV, infadp.py:167,185)."""

    def __init__(self, value: GopsMlp, batch: int, device: Optional[torch.device] = None):
        self.mlp, self.batch = value, batch
        nbytes = lib().gops_value_workspace_bytes(C.byref(value), batch)
        if nbytes == 0:
            raise RuntimeError("gops_value_workspace_bytes: unsupported value network for the HIP path")
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def forward(self, obs: torch.Tensor) -> torch.Tensor:
        v = torch.empty(self.batch, dtype=torch.float32, device=self.device)
        check(lib().gops_value_forward(C.byref(self.mlp), self.batch, _ptr(obs), _ptr(v),
                                       self.workspace.data_ptr(), self.workspace.numel(), _stream()),
              "gops_value_forward")
        return v

    def backward(self, obs: torch.Tensor, grad_v: torch.Tensor, grad_w, grad_b, tail: Optional["GopsUpdateTail"] = None):
        """`tail` (ABI v12, `gops_value_backward_update`): Adam step (+ Polyak step of the target) folded into the launch that forms the gradients."""
        g = make_mlp_grad(grad_w, grad_b)
        if tail is not None:
            check(lib().gops_value_backward_update(C.byref(self.mlp), self.batch, _ptr(obs), _ptr(grad_v), C.byref(g), C.byref(tail),
                                                   self.workspace.data_ptr(), self.workspace.numel(), _stream()),
                  "gops_value_backward_update")
            return
        check(lib().gops_value_backward(C.byref(self.mlp), self.batch, _ptr(obs), _ptr(grad_v), C.byref(g),
                                        self.workspace.data_ptr(), self.workspace.numel(), _stream()),
              "gops_value_backward")


class MlpNet:
    """Batch evaluation of an MLP with an output layer of any width and its backward into the parameters
    (`gops_mlp_forward / _backward`): FiniteHorizonFullPolicy's single evaluation in FHADP2."""

    def __init__(self, mlp: GopsMlp, batch: int, device: Optional[torch.device] = None):
        self.mlp, self.batch = mlp, batch
        self.out_dim = int(mlp.sizes[mlp.n_layers])
        nbytes = lib().gops_mlp_workspace_bytes(C.byref(mlp), batch)
        if nbytes == 0:
            raise RuntimeError("gops_mlp_workspace_bytes: unsupported network for the HIP path")
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = torch.empty(self.batch, self.out_dim, dtype=torch.float32, device=self.device)
        check(lib().gops_mlp_forward(C.byref(self.mlp), self.batch, _ptr(x), _ptr(y), self.workspace.data_ptr(),
                                     self.workspace.numel(), _stream()), "gops_mlp_forward")
        return y

    def backward(self, x: torch.Tensor, grad_y: torch.Tensor, grad_w, grad_b):
        g = make_mlp_grad(grad_w, grad_b)
        check(lib().gops_mlp_backward(C.byref(self.mlp), self.batch, _ptr(x), _ptr(grad_y), C.byref(g),
                                      self.workspace.data_ptr(), self.workspace.numel(), _stream()), "gops_mlp_backward")

    def backward_x(self, x: torch.Tensor, grad_y: torch.Tensor, grad_w=None, grad_b=None) -> torch.Tensor:
        """`gops_mlp_backward_x`: returns d(loss)/d(x); parameter gradients too when grad_w / grad_b are given."""
        g = make_mlp_grad(grad_w, grad_b) if grad_w is not None else None
        gx = torch.empty(self.batch, int(self.mlp.sizes[0]), dtype=torch.float32, device=self.device)
        check(lib().gops_mlp_backward_x(C.byref(self.mlp), self.batch, _ptr(x), _ptr(grad_y), C.byref(g) if g is not None else None,
                                        _ptr(gx), self.workspace.data_ptr(), self.workspace.numel(), _stream()),
              "gops_mlp_backward_x")
        return gx


def env_step(env: GopsEnv, obs, action, done, info: Optional[Dict[str, torch.Tensor]] = None):
    """One wrapped env-model step on the GPU (gops_env_step).  `info["ref_appended"]` [B, 4] (optional): the reference
    point this step appends, from the caller (bit-parity mode, GopsStepIO.ref_appended)."""
    B = obs.shape[0]
    io = GopsStepIO()
    nobs, rew, ndone = torch.empty_like(obs), torch.empty(B, device=obs.device), torch.empty(B, device=obs.device)
    io.obs, io.action, io.done = _ptr(obs), _ptr(action), _ptr(done)
    io.next_obs, io.reward, io.next_done = _ptr(nobs), _ptr(rew), _ptr(ndone)
    ninfo = {}
    if env.kind in (ENV_VEH, ENV_VEH_SURR, ENV_VEH2DOF):
        for k in ("state", "ref_points", "path_num", "u_num", "ref_time"):
            setattr(io, k, _ptr(info[k]))
        ninfo = dict(state=torch.empty_like(info["state"]), ref_points=torch.empty_like(info["ref_points"]),
                     ref_time=torch.empty_like(info["ref_time"]), path_num=info["path_num"], u_num=info["u_num"])
        io.next_state, io.next_ref_points = _ptr(ninfo["state"]), _ptr(ninfo["ref_points"])
        io.next_ref_time = _ptr(ninfo["ref_time"])
        io.ref_appended = _ptr(info.get("ref_appended"))
    if env.kind == ENV_MOBILEROBOT:   # info["noise"] [B, 2]: this step's obstacle draws (drawn here unless handed in)
        noise = (info or {}).get("noise")
        if noise is None:
            noise = mobilerobot_noise((B, 2), obs.device)
        io.noise = _ptr(noise)
    if has_constraints(env) and env.n_surr == 0:
        ninfo["constraint"] = torch.empty(B, env.n_constraint, dtype=torch.float32, device=obs.device)
        io.constraint = _ptr(ninfo["constraint"])
    elif env.kind == ENV_VEH_SURR:
        ninfo["surr_state"] = torch.empty_like(info["surr_state"])
        ninfo["constraint"] = torch.empty(B, env.n_constraint, dtype=torch.float32, device=obs.device)
        io.surr_state, io.next_surr_state = _ptr(info["surr_state"]), _ptr(ninfo["surr_state"])
        io.constraint = _ptr(ninfo["constraint"])
    check(lib().gops_env_step(C.byref(env), B, C.byref(io), _stream()), "gops_env_step")
    return nobs, rew, ndone, ninfo


def env_constraint(env: GopsEnv, obs, info: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """model.get_constraint(obs, info) on the GPU (gops_env_constraint): [B, n_constraint]."""
    B = obs.shape[0]
    io = GopsStepIO()
    out = torch.empty(B, env.n_constraint, dtype=torch.float32, device=obs.device)
    io.obs, io.constraint = _ptr(obs), _ptr(out)
    if info:
        io.state, io.surr_state = _ptr(info.get("state")), _ptr(info.get("surr_state"))
    check(lib().gops_env_constraint(C.byref(env), B, C.byref(io), _stream()), "gops_env_constraint")
    return out


class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam (default hyper-parameters' semantics) whose `step()` is ONE HIP launch over all
    parameters of a group (`gops_adam_step`).  Same `param_groups` / `state` layout as torch's Adam
    (`step`, `exp_avg`, `exp_avg_sq`), so lr schedulers and optimizer checkpoints are interchangeable.
    The learning rate and step count the kernel uses live in device memory, which makes `step()`
    capturable in a HIP graph: a graph owner calls `sync_hyper()` before and `advance()` after each
    replay.  Parameters must be fp32 CUDA tensors when `step()` is called."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._dev = {}   # group index -> {"state": GopsAdamState device tensors (6 x 8 bytes) per chunk, "lr", "step"}
        # factor applied to every gradient element inside the kernel: 1/N after a SUM all-reduce over N replicas
        # (trainer/grad_sync.py) - the mean costs no separate pass over the gradients
        self.grad_scale = 1.0

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._dev = {}   # device mirrors are rebuilt from the loaded host state

    def _prepare(self, gi, group):
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return None, None
        steps = set()
        for p in ps:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
            elif st["exp_avg"].device != p.device:   # parameters were moved after the state was made
                st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].to(p.device), st["exp_avg_sq"].to(p.device)
            steps.add(int(st["step"]))
        assert len(steps) == 1, "parameters of one group must share the step count"
        step = steps.pop()
        nchunk = (len(ps) + ADAM_MAX - 1) // ADAM_MAX
        dev = self._dev.get(gi)
        if dev is None or len(dev["state"]) != nchunk or dev["state"][0].device != ps[0].device:
            dev = self._dev[gi] = {"state": [torch.zeros(6, dtype=torch.int64, device=ps[0].device)
                                             for _ in range(nchunk)], "lr": None, "step": None, "gs": None}
        if dev["step"] != step:       # first use, after load_state_dict, or an external edit of state
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("HipAdam: device step count out of date while capturing a graph")
            b1, b2 = group["betas"]
            for t in dev["state"]:   # GopsAdamState: lr, step, beta1^step, beta2^step, (ticket, skipped_nonfinite), grad_scale
                t[1] = step
                t.view(torch.float64)[2] = float(b1) ** step
                t.view(torch.float64)[3] = float(b2) ** step
                t.view(torch.int32)[8] = 0   # ticket (the skipped-element counter next to it keeps counting)
            dev["step"] = step
        self._sync_lr(dev, group)
        return ps, dev

    def _sync_lr(self, dev, group):
        lr = float(group["lr"])
        if dev["lr"] != lr:
            for t in dev["state"]:
                t.view(torch.float64)[0] = lr
            dev["lr"] = lr
        gs = float(self.grad_scale)
        if dev.get("gs") != gs:
            for t in dev["state"]:
                t.view(torch.float64)[5] = gs
            dev["gs"] = gs

    def skipped_nonfinite(self) -> int:
        """Gradient elements that were not finite and therefore took no step since the device state was created
        (`GopsAdamState.skipped_nonfinite`, ABI v13: parameter, moments and a fused Polyak target of such an element stay
        untouched).  Reads device memory: one host sync."""
        return sum(int(t.view(torch.int32)[9].item()) & 0xFFFFFFFF for dev in self._dev.values() for t in dev["state"])

    def sync_hyper(self):
        """Push a changed `lr` (schedulers) to the device copies; call before replaying a graph."""
        for gi, group in enumerate(self.param_groups):
            if gi in self._dev:
                self._sync_lr(self._dev[gi], group)

    def storage_signature(self):
        """Device addresses a captured `step()` has baked in besides parameters and gradients: the Adam moments
        and the device-resident hyper-parameter blocks.  `load_state_dict` (new moment tensors, `_dev` dropped)
        changes it, which makes graph owners re-capture instead of replaying into freed memory."""
        sig = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                st = self.state.get(p, {})
                if len(st):
                    sig.append((st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()))
            if gi in self._dev:
                sig.append(tuple(t.data_ptr() for t in self._dev[gi]["state"]))
        return tuple(sig)

    def resync_device_state(self):
        """Forget the device mirrors of (lr, step, beta powers): the next `step()` re-pushes them from the host
        state.  Used after an aborted graph capture, whose host-side step bookkeeping ran without the kernel."""
        for dev in self._dev.values():
            dev["step"] = dev["lr"] = dev["gs"] = None

    def advance(self, n: int = 1):
        """Account for `n` graph replays of `step()` in the host-side step counters."""
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p in self.state and len(self.state[p]):
                    self.state[p]["step"] = int(self.state[p]["step"]) + n
            if gi in self._dev and self._dev[gi]["step"] is not None:
                self._dev[gi]["step"] += n

    def begin_fused(self):
        """The table / device state / hyper-parameters of THIS step for a caller that folds it into another launch
        (`gops_rollout_backward_update`): `(GopsAdamTensors, state pointer, beta1, beta2, eps, keep-alive)` - or None when the
        optimizer is not one group whose parameters fit one table (the caller then calls `step()` as usual).  Exactly the
        preparation `step()` does; the caller reports the enqueued launch with `end_fused()`."""
        if len(self.param_groups) != 1:
            return None
        group = self.param_groups[0]
        ps, dev = self._prepare(0, group)
        if ps is None or len(ps) > ADAM_MAX or len(ps) != len(group["params"]):
            return None
        t = GopsAdamTensors()
        t.n = len(ps)
        keep = []
        for i, p in enumerate(ps):
            st = self.state[p]
            if not p.grad.is_contiguous():
                return None
            keep.append(p.grad)
            t.numel[i], t.param[i], t.grad[i] = p.numel(), _ptr(p.data), _ptr(p.grad)
            t.exp_avg[i], t.exp_avg_sq[i] = _ptr(st["exp_avg"]), _ptr(st["exp_avg_sq"])
        b1, b2 = group["betas"]
        self._fused = (ps, dev)
        return t, dev["state"][0].data_ptr(), float(b1), float(b2), float(group["eps"]), keep

    def end_fused(self):
        """Host-side step bookkeeping of the launch `begin_fused` prepared (what `step()` does after its launch)."""
        ps, dev = self._fused
        self._fused = None
        for p in ps:
            self.state[p]["step"] = int(self.state[p]["step"]) + 1
        dev["step"] += 1

    @torch.no_grad()
    def step(self, closure=None):
        for gi, group in enumerate(self.param_groups):
            ps, dev = self._prepare(gi, group)
            if ps is None:
                continue
            b1, b2 = group["betas"]
            for ci, i0 in enumerate(range(0, len(ps), ADAM_MAX)):
                chunk = ps[i0:i0 + ADAM_MAX]
                t = GopsAdamTensors()
                t.n = len(chunk)
                keep = []
                for i, p in enumerate(chunk):
                    st = self.state[p]
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    keep.append(g)
                    t.numel[i], t.param[i], t.grad[i] = p.numel(), _ptr(p.data), _ptr(g)
                    t.exp_avg[i], t.exp_avg_sq[i] = _ptr(st["exp_avg"]), _ptr(st["exp_avg_sq"])
                check(lib().gops_adam_step(C.byref(t), dev["state"][ci].data_ptr(), b1, b2, group["eps"], _stream()),
                      "gops_adam_step")
            for p in ps:
                self.state[p]["step"] = int(self.state[p]["step"]) + 1
            dev["step"] += 1


LOSS_STATS_FLOATS = 260   # include/gops_hip.h: GOPS_LOSS_STATS_FLOATS


class LossStats:
    """Device scalars of a batch in one launch (`gops_value_loss` / `gops_mean_loss`): `stats[0]`, `stats[1]` are the results,
    the rest of the buffer is the kernels' scratch (zero before the first call, left zero by every call)."""

    def __init__(self, device):
        self.buf = torch.zeros(LOSS_STATS_FLOATS, dtype=torch.float32, device=device)

    def value_loss(self, v: torch.Tensor, target: torch.Tensor, grad: Optional[torch.Tensor] = None) -> torch.Tensor:
        """-> [mean((v - target)^2), mean(v)]; `grad` <- (2 / n) (v - target) (gops/algorithm/infadp.py:172-173)."""
        n = v.numel()
        check(lib().gops_value_loss(_ptr(v), _ptr(target), n, _ptr(grad), self.buf.data_ptr(), _stream()), "gops_value_loss")
        return self.buf[:2]

    def mean_loss(self, x: torch.Tensor, scale: float = -1.0) -> torch.Tensor:
        """-> [scale * mean(x), mean(x)] (`-v_pi.mean()`: infadp.py:213, fhadp.py:123)."""
        check(lib().gops_mean_loss(_ptr(x), x.numel(), float(scale), self.buf.data_ptr(), _stream()), "gops_mean_loss")
        return self.buf[:2]


def make_update_tail(fused_adam=None, mean_of: Optional[torch.Tensor] = None, mean_scale: float = -1.0,
                     stats: Optional["LossStats"] = None, polyak: Optional["PolyakUpdater"] = None, tau: float = 0.0) -> "GopsUpdateTail":
    """`GopsUpdateTail` for `Rollout.backward(..., tail=)` / `ValueNet.backward(..., tail=)`: `fused_adam` = what
    `HipAdam.begin_fused()` returned (or None), `mean_of` / `stats`: the values whose mean lands in `stats.buf[:2]` as `gops_mean_loss`
    would leave it, `polyak` / `tau`: the target-network averaging of the stepped network (a one-table `PolyakUpdater`).  The
    returned struct keeps the tensors it points to alive."""
    t = GopsUpdateTail()
    keep = []
    if fused_adam is not None:
        table, state_ptr, b1, b2, eps, kp = fused_adam
        t.adam = C.pointer(table)
        t.adam_state, t.beta1, t.beta2, t.eps = state_ptr, b1, b2, eps
        keep += [table, kp]
    if mean_of is not None:
        t.mean_x, t.mean_n, t.mean_scale, t.mean_stats = _ptr(mean_of), mean_of.numel(), float(mean_scale), stats.buf.data_ptr()
        keep += [mean_of, stats]
    if polyak is not None:
        assert fused_adam is not None and len(polyak.tables) == 1
        t.polyak, t.polyak_tau = C.pointer(polyak.tables[0]), float(tau)
        keep.append(polyak)
    t._keep = keep
    return t


class PolyakUpdater:
    """target <- (1 - tau) target + tau online for all tensors of a network in one launch (`gops_polyak_update`; chunks of
    GOPS_ADAM_MAX_TENSORS), the same two roundings per element as the reference's `mul_` / `add_` passes (infadp.py:124-133)."""

    def __init__(self, target_params, online_params):
        tp, op = list(target_params), list(online_params)
        assert len(tp) == len(op)
        self._keep = (tp, op)
        self._sig = tuple((t.data_ptr(), o.data_ptr(), t.numel()) for t, o in zip(tp, op))
        self.tables = []
        for c0 in range(0, len(tp), ADAM_MAX):
            t = GopsAdamTensors()
            chunk = list(zip(tp[c0:c0 + ADAM_MAX], op[c0:c0 + ADAM_MAX]))
            t.n = len(chunk)
            for i, (a, b) in enumerate(chunk):
                assert a.shape == b.shape
                t.numel[i] = a.numel()
                t.param[i], t.grad[i] = _ptr(a.data), _ptr(b.data)
            self.tables.append(t)

    def matches(self, target_params, online_params) -> bool:
        return self._sig == tuple((t.data_ptr(), o.data_ptr(), t.numel()) for t, o in zip(target_params, online_params))

    def step(self, tau: float):
        for t in self.tables:
            check(lib().gops_polyak_update(C.byref(t), float(tau), _stream()), "gops_polyak_update")


def profile_enable(on: bool):
    lib().gops_profile_enable(int(on))


def profile_reset():
    lib().gops_profile_reset()


def profile_read(kernel_id: int):
    ms, n = C.c_double(0.0), C.c_int64(0)
    check(lib().gops_profile_read(kernel_id, C.byref(ms), C.byref(n)), "gops_profile_read")
    return ms.value, n.value
