"""CPU-only checks of the host layer: C-ABI library exports, factories/registries, error
behaviour, state_dict layout and the N>1 gradient exchange (gloo, world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fhadp_kwargs(**over):
    kw = dict(algorithm="FHADP", trainer="on_serial_trainer", seed=0, env_id="pyth_idpendulum", pre_horizon=10,
              obsv_dim=6, action_dim=1, action_type="continu", action_high_limit=np.ones(1, np.float32),
              action_low_limit=-np.ones(1, np.float32), policy_func_type="MLP",
              policy_func_name="FiniteHorizonPolicy", policy_hidden_sizes=[64, 64],
              policy_hidden_activation="gelu", policy_act_distribution="default", policy_learning_rate=1e-3,
              use_gpu=False)
    kw.update(over)
    return kw


def test_library_exports_every_declared_symbol():
    """Every function declared in include/gops_hip.h is exported by the built library - and nothing else (the library is
    built with -fvisibility=hidden: no mangled internals beside the C ABI)."""
    import subprocess
    from gops_amd import hip_backend as hb
    header = open(os.path.join(ROOT, "include", "gops_hip.h")).read()
    declared = set(re.findall(r"\b(gops_[a-z_]+)\s*\(", header))
    assert declared == set(hb.EXPORTED_SYMBOLS)
    assert os.path.exists(hb.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(hb.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert hb.lib().gops_hip_version() == int(re.search(r"#define GOPS_HIP_ABI_VERSION (\d+)", header).group(1))
    out = subprocess.run(["nm", "-D", "--defined-only", hb.LIB_PATH], capture_output=True, text=True, check=True).stdout
    extra = [l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1] not in declared]
    assert extra == [], extra


def test_device_code_has_no_packed_fp32_instructions():
    """DESIGN_LOG.md, round 4 (round 4): on gfx950 a v_pk_fma_f32 that consumes the result of a v_pk_mul_f32 issued two slots
    earlier reads zeros in lanes 48..63 when another wave of the SIMD streams MFMAs (stand-alone reproducer:
    tools/microbench/pk_hazard.hip) - the cause of the run-to-run non-determinism of the round-3 streamed plane-split kernels.
    The library is therefore built with the `packed-fp32-ops` subtarget feature off (csrc/Makefile NOPK); this test disassembles
    every embedded gfx950 code object and fails if the flag got lost."""
    import sys
    from gops_amd import hip_backend as hb
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from scan_code_objects import scan
    pk, mfma = r"v_pk_(mul|add|fma)_f32", r"v_mfma_f32_16x16x32_(bf16|f16)"
    counts, kernels = scan(hb.LIB_PATH, [pk, mfma])
    assert kernels > 100, kernels      # the scan saw the rollout kernels ...
    assert counts[mfma] > 1000         # ... and their disassembly
    assert counts[pk] == 0, counts

def test_ctypes_mirror_matches_the_header_layout(tmp_path):
    """`gops_amd/hip_backend.py` restates every struct of include/gops_hip.h by hand (ctypes).  This compiles a C program against
    the header (plain C: the boundary is a C ABI) that prints sizeof of every struct and offsetof of every member, and holds
    the ctypes classes to it - a field added on one side only, or moved into what used to be padding, fails here."""
    from gops_amd import hip_backend as hb
    pairs = {"GopsMlp": hb.GopsMlp, "GopsMlpGrad": hb.GopsMlpGrad, "GopsEnv": hb.GopsEnv, "GopsRolloutDesc": hb.GopsRolloutDesc,
             "GopsRolloutIn": hb.GopsRolloutIn, "GopsRolloutOut": hb.GopsRolloutOut, "GopsRolloutAdjoint": hb.GopsRolloutAdjoint,
             "GopsStepIO": hb.GopsStepIO, "GopsAdamTensors": hb.GopsAdamTensors, "GopsUpdateTail": hb.GopsUpdateTail}
    header = open(os.path.join(ROOT, "include", "gops_hip.h")).read()
    declared = set(re.findall(r"typedef struct (Gops[A-Za-z]+) \{", header))
    assert declared - {"GopsAdamState"} == set(pairs), "a struct of the header has no ctypes mirror (or the reverse)"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "gops_hip.h"', "int main(void) {"]
    for name, cls in pairs.items():
        lines.append(f'    printf("{name} %zu\\n", sizeof({name}));')
        for field, _ in cls._fields_:
            lines.append(f'    printf("{name}.{field} %zu\\n", offsetof({name}, {field}));')
    lines.append('    printf("GopsAdamState %zu\\n", sizeof(GopsAdamState));')
    lines += ["    return 0;", "}"]
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text("\n".join(lines))
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for name, cls in pairs.items():
        assert int(got[name]) == ctypes.sizeof(cls), (name, got[name], ctypes.sizeof(cls))
        for field, _ in cls._fields_:
            assert int(got[f"{name}.{field}"]) == getattr(cls, field).offset, (name, field, got[f"{name}.{field}"], getattr(cls, field).offset)
    assert int(got["GopsAdamState"]) == 48   # HipAdam keeps it as 6 x int64 of device memory


def test_integration_md_stub_matches_the_current_abi():
    """INTEGRATION.md section 2 shows the ctypes stub a GOPS maintainer would add: its `GopsMlp` must be the layout of the
    CURRENT header (v10 put `variant_flags` where v9 had padding - a stale copy would still run and silently pass garbage)."""
    from gops_amd import hip_backend as hb
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"(class GopsMlp\(C\.Structure\):\n(?:    .*\n)+)", text)
    assert m, "INTEGRATION.md lost its GopsMlp stub"
    ns = {"C": ctypes}
    exec(m.group(1), ns)
    stub = ns["GopsMlp"]
    assert [f[0] for f in stub._fields_] == [f[0] for f in hb.GopsMlp._fields_]
    assert ctypes.sizeof(stub) == ctypes.sizeof(hb.GopsMlp)
    for name, _ in stub._fields_:
        assert getattr(stub, name).offset == getattr(hb.GopsMlp, name).offset, name
    header = open(os.path.join(ROOT, "include", "gops_hip.h")).read()
    abi = re.search(r"#define GOPS_HIP_ABI_VERSION (\d+)", header).group(1)
    assert f"gops_hip_version() == {abi}" in text, "the stub asserts another ABI version than the header's"


def test_precision_guard_schedule_and_decision():
    """algorithm/base.py PrecisionGuard without a GPU: which gradients are checked, when it trips, that it is sticky, and the
    flags it hands the launches."""
    from gops_amd import hip_backend as hb
    from gops_amd.algorithm.base import PrecisionGuard
    g = PrecisionGuard(interval=4, threshold=5e-5)
    assert [g.due() for _ in range(9)] == [True, False, False, True, False, False, False, True, False]   # gradients 1, 4, 8
    base = torch.arange(1.0, 101.0)
    seen = []

    def close(flags):
        seen.append(flags)
        return base * (1 + (1e-6 if flags else 0.0))
    d = g.check(close)
    assert seen == [hb.DEFAULT_VARIANT_FLAGS, hb.DEFAULT_VARIANT_FLAGS | PrecisionGuard.exact_rollout_flags()]
    assert 0.5e-6 < d < 2e-6 and not g.exact and g.flags() == hb.DEFAULT_VARIANT_FLAGS
    with pytest.warns(UserWarning, match="exact-fp32 rollout kernels"):
        d = g.check(lambda flags: base * (1 + (3e-4 if flags else 0.0)))
    assert 2e-4 < d < 4e-4 and g.exact
    assert g.flags() & PrecisionGuard.exact_rollout_flags() == PrecisionGuard.exact_rollout_flags()
    assert not any(g.due() for _ in range(20))   # sticky: nothing left to decide
    with pytest.warns(UserWarning):
        g2 = PrecisionGuard(interval=1, threshold=5e-5)
        g2.check(lambda flags: base * float("nan") if flags == 0 else base)   # a NaN gradient counts as exceeded
    assert g2.exact
    off = PrecisionGuard(interval=0)
    assert not any(off.due() for _ in range(5))
    # nothing to guard for nets that never reach a plane-split kernel (hidden width != 256) or for a stochastic model
    from gops_amd.create_pkg.create_alg import create_alg
    narrow = create_alg(**_fhadp_kwargs()).networks.policy
    wide = create_alg(**_fhadp_kwargs(policy_hidden_sizes=[256, 256])).networks.policy
    assert not PrecisionGuard.applies_to(narrow) and PrecisionGuard.applies_to(wide) and PrecisionGuard.applies_to(narrow, wide)
    assert not PrecisionGuard.applies_to(wide, env_kind=hb.ENV_MOBILEROBOT)
    # the exact-rollout flags leave the weight-gradient GEMM alone
    f = PrecisionGuard.exact_rollout_flags()
    assert f & (hb.VF_NO_STREAMED_SPLIT_BWD | hb.VF_DW_F32 | hb.VF_DW_EXACT | hb.VF_STREAMED_FP32) == 0


def test_workspace_query_and_rejections_need_no_gpu():
    from gops_amd import hip_backend as hb
    d = hb.GopsRolloutDesc()
    assert hb.lib().gops_rollout_workspace_bytes(ctypes.byref(d)) == 0     # empty descriptor rejected
    from gops_amd.create_pkg.create_env_model import create_env_model
    env = create_env_model("pyth_veh3dofconti", pre_horizon=30).hip_env()
    d.batch, d.horizon, d.finite_horizon, d.need_grad, d.gamma, d.env = 4096, 30, 1, 1, 1.0, env
    m = hb.GopsMlp()
    m.n_layers = 3
    for i, s in enumerate([127, 256, 256, 2]):
        m.sizes[i] = s
    for j in range(3):
        m.weight[j] = m.bias[j] = 1   # non-null placeholders: the size query never dereferences
    m.hidden_act = hb.ACT_IDS["elu"]
    d.policy = m
    nbytes = hb.lib().gops_rollout_workspace_bytes(ctypes.byref(d))
    stash = 4096 * 30 * 4 * (128 + 2 * 256 + 2 * 256)   # X + H1,H2 + D1,D2
    assert stash < nbytes < 2 * stash
    m.sizes[1] = 250                                     # hidden width not a multiple of 16
    d.policy = m
    assert hb.lib().gops_rollout_workspace_bytes(ctypes.byref(d)) == 0


def test_update_tail_entry_points_reject_bad_arguments_without_a_gpu():
    """ABI v12: `gops_rollout_backward_update` / `gops_value_backward_update` check their arguments before anything is launched - a
    missing tail, a missing descriptor or gradient table is GOPS_ERR_BAD_ARG (-1), never a crash and never a silent plain backward."""
    from gops_amd import hip_backend as hb
    lib = hb.lib()
    d, i, g, t, m = hb.GopsRolloutDesc(), hb.GopsRolloutIn(), hb.GopsMlpGrad(), hb.GopsUpdateTail(), hb.GopsMlp()
    assert lib.gops_rollout_backward_update(ctypes.byref(d), ctypes.byref(i), None, ctypes.byref(g), None, None, 0, None) < 0
    assert lib.gops_rollout_backward_update(None, ctypes.byref(i), None, ctypes.byref(g), ctypes.byref(t), None, 0, None) < 0
    assert lib.gops_rollout_backward_update(ctypes.byref(d), ctypes.byref(i), None, ctypes.byref(g), ctypes.byref(t), None, 0, None) < 0   # (no grad_v, empty descriptor)
    assert lib.gops_value_backward_update(ctypes.byref(m), 64, None, None, ctypes.byref(g), ctypes.byref(t), None, 0, None) < 0
    assert lib.gops_value_backward_update(ctypes.byref(m), 64, 1, 1, ctypes.byref(g), None, None, 0, None) < 0


def test_variant_selection_is_part_of_the_description_not_of_the_process():
    """ABI v10: kernel variants are chosen by GopsRolloutDesc.variant_flags - two callers in one process can choose
    differently (the library keeps no global state) - and the library source reads the process environment in ONE
    place only (the debug override, read once at load)."""
    import glob
    from gops_amd import hip_backend as hb
    from gops_amd.create_pkg.create_env_model import create_env_model
    d = hb.GopsRolloutDesc()
    d.batch, d.horizon, d.finite_horizon, d.need_grad, d.gamma = 4096, 30, 1, 1, 1.0
    d.env = create_env_model("pyth_veh3dofconti", pre_horizon=30).hip_env()
    m = hb.GopsMlp()
    m.n_layers = 3
    for i, s in enumerate([127, 256, 256, 2]):
        m.sizes[i] = s
    for j in range(3):
        m.weight[j] = m.bias[j] = 1
    m.hidden_act = hb.ACT_IDS["elu"]
    d.policy = m
    variant = lambda flags: (setattr(d, "variant_flags", flags), hb.lib().gops_rollout_variant(ctypes.byref(d)))[1]
    assert variant(0) == 1                                    # register-stationary plane-split kernels (the headline launch)
    assert variant(hb.VF_NO_STATIONARY_SPLIT) == 2            # register-stationary fp32-MFMA kernels
    assert variant(hb.VF_STREAMED_FP32) == 0                  # plain streamed fp32 kernels
    assert variant(0) == 1                                    # ... and nothing lingered
    m.n_layers = 4                                            # three hidden layers: the streamed plane-split kernels
    for i, s in enumerate([47, 256, 256, 256, 2]):            # (P = 10: two of its workgroups fit a CU's LDS)
        m.sizes[i] = s
    m.weight[3] = m.bias[3] = 1
    d.policy = m
    d.env = create_env_model("pyth_veh3dofconti", pre_horizon=10).hip_env()
    assert variant(0) == 4 and variant(hb.VF_NO_STREAMED_SPLIT_FWD) == 2 and variant(hb.VF_STREAMED_FP32) == 0
    # relu nets with a tail value net (INFADP, cfg3): still the streamed plane-split kernels - only the tail value net of a launch
    # that keeps a gradient is evaluated with exact fp32 products (round 4; round 3 kept the whole launch on the fp32 kernels)
    m.hidden_act = hb.ACT_IDS["relu"]
    m.sizes[0] = 46                                           # (infinite horizon: no time column)
    d.policy = m
    v = hb.GopsMlp()
    v.n_layers = 4
    for i, s in enumerate([46, 256, 256, 256, 1]):
        v.sizes[i] = s
    for j in range(4):
        v.weight[j] = v.bias[j] = 1
    v.hidden_act = hb.ACT_IDS["relu"]
    d.value, d.tail_value, d.finite_horizon, d.batch, d.horizon = v, 1, 0, 8192, 10
    for ng in (1, 0):
        d.need_grad = ng
        assert variant(0) == 4, ng
    n_getenv = sum(open(f).read().count("getenv(") for f in glob.glob(os.path.join(ROOT, "gops_amd", "csrc", "*.h*")))
    assert n_getenv <= 3, n_getenv


def test_registries_and_error_behaviour():
    from gops_amd.create_pkg import create_alg, create_apprfunc, create_env_model, create_trainer
    assert set(create_alg.registry) == {"FHADP", "FHADP2", "FHADPExterior", "FHADPInterior", "FHADPLagrangian", "INFADP", "MAC", "MPG", "SPIL"}
    assert {"mlp_DetermPolicy", "mlp_FiniteHorizonPolicy", "mlp_StateValue"} <= set(create_apprfunc.registry)
    assert {"pyth_lq_model", "pyth_idpendulum_model", "pyth_veh3dofconti_model", "pyth_veh3dofconti_surrcstr_model",
            "pyth_veh3dofconti_detour_model", "pyth_veh3dofconti_surrcstr_penalty_model", "gym_cartpoleconti_model",
            "gym_pendulum_model", "pyth_veh2dofconti_model", "pyth_veh3dofconti_errcstr_model",
            "pyth_veh2dofconti_errcstr_model"} <= set(create_env_model.registry)
    assert {"on_serial_trainer", "on_sync_trainer", "off_serial_trainer", "off_sync_trainer",
            "off_async_trainer"} <= set(create_trainer.registry)
    with pytest.raises(KeyError, match="No registered algorithm with id"):
        create_alg.create_alg(algorithm="NOPE")
    with pytest.raises(KeyError, match="No registered env with id"):
        create_env_model.create_env_model("pyth_nope")
    with pytest.raises(KeyError, match="No registered apprfunc with id"):
        create_apprfunc.create_apprfunc(apprfunc="MLP", name="Nope")
    with pytest.raises(KeyError, match="No registered trainer with id"):
        create_trainer.create_trainer(None, None, None, None, trainer="nope_trainer")
    # wrapper options outside the kernels' contract are refused, never silently ignored
    with pytest.raises(RuntimeError, match="repeat_num"):
        create_env_model.create_env_model("pyth_veh3dofconti", repeat_num=2)   # the reference wrapper does not advance `info`
    with pytest.raises(RuntimeError, match="repeat_num must be"):
        create_env_model.create_env_model("pyth_lq", repeat_num=9)
    with pytest.raises(RuntimeError, match="mask_at_done"):
        create_env_model.create_env_model("pyth_veh3dofconti_surrcstr", mask_at_done=False)
    assert create_env_model.create_env_model("pyth_lq", mask_at_done=False).hip_env().no_mask_at_done == 1
    m = create_env_model.create_env_model("pyth_idpendulum", repeat_num=3, sum_reward=False)
    assert m.hip_env().repeat_num == 3 and m.hip_env().repeat_last_reward == 1


def test_reference_trajectory_constants_fold_like_the_reference():
    """`ref_constants` (GopsEnv.ref_c): the caller's path_para / u_para update the default set per profile (unknown profile
    names raise like the reference's dict update), and every derived constant is folded in double where the reference
    multiplies Python scalars."""
    import math
    from gops_amd.env.env_ocp.resources.ref_traj_params import merged, ref_constants
    c = ref_constants()
    w = 2 * math.pi / 10
    assert len(c) == 24 and c[:7] == [-1.0 / w, w, 0.0, 5.0, 1.0 / w, 1.0, 5.0]
    assert c[7:10] == [1.5, w, 0.0] and c[10:18] == [5.0, 9.0, 14.0, 18.0, 0.0, 3.5, 0.875, -0.875]
    assert c[18:23] == [10.0, 0.6, -0.6, 5.0, 100.0]
    c2 = ref_constants({"double_lane": {"y2": 3.0, "t2": 8.5}}, {"sine": {"phi": 0.2, "A": 1.5}})
    assert c2[16] == (3.0 - 0.0) / (8.5 - 5.0) and c2[17] == (0.0 - 3.0) / (18.0 - 14.0)
    assert c2[0] == -1.5 / w and c2[4] == 1.5 / w * math.cos(0.2) and c2[7:10] == c[7:10]
    path, speed = merged({"circle": {"r": 80.0}}, None)
    assert path["circle"]["r"] == 80.0 and path["sine"]["A"] == 1.5 and speed["constant"]["u"] == 5.0
    with pytest.raises(KeyError):
        ref_constants({"spiral": {"r": 1.0}})
    from gops_amd.create_pkg import create_env_model
    m = create_env_model.create_env_model("pyth_veh3dofconti", pre_horizon=10, path_para={"circle": {"r": 80.0}})
    e = m.hip_env()
    assert e.ref_custom == 1 and e.ref_c[22] == 80.0
    assert create_env_model.create_env_model("pyth_veh3dofconti", pre_horizon=10).hip_env().ref_custom == 0


def test_create_alg_hands_out_actor_handles_for_the_ray_trainers():
    """For off_sync / off_async the reference's create_alg returns a list of Ray actor handles (create_alg.py:87-93) and
    the example scripts talk to them through `.remote(...)`; here the list holds this rank's replica behind the same
    call syntax, and the trainers unwrap it."""
    from gops_amd.create_pkg.create_alg import LocalActor, create_alg
    algs = create_alg(**dict(_fhadp_kwargs(), trainer="off_async_trainer"))
    assert isinstance(algs, list) and len(algs) == 1 and isinstance(algs[0], LocalActor)
    for a in algs:
        a.set_parameters.remote({"gamma": 0.95})          # the scripts' idiom
    assert algs[0].unwrap().gamma == 0.95 and algs[0].get_parameters.remote()["gamma"] == 0.95
    assert algs[0].networks is algs[0].unwrap().networks   # plain attributes pass through
    assert not isinstance(create_alg(**dict(_fhadp_kwargs(), trainer="on_sync_trainer")), list)   # (create_alg.py:80-85)


def test_state_dict_layout_and_parameter_api():
    from gops_amd.create_pkg.create_alg import create_alg
    alg = create_alg(**_fhadp_kwargs())
    assert list(alg.state_dict()) == ["policy.act_high_lim", "policy.act_low_lim", "policy.pi.0.weight",
                                      "policy.pi.0.bias", "policy.pi.2.weight", "policy.pi.2.bias",
                                      "policy.pi.4.weight", "policy.pi.4.bias"]
    assert alg.networks.policy.pi[0].weight.shape == (64, 7)       # obs + virtual time column
    alg.set_parameters({"gamma": 0.9, "pre_horizon": 5})
    assert alg.get_parameters() == {"pre_horizon": 5, "gamma": 0.9}
    with pytest.raises(RuntimeError):
        alg.set_parameters({"tau": 0.1})
    a = alg.networks.policy(torch.zeros(3, 6), 4)
    assert a.shape == (3, 1) and a.abs().max() <= 1
    inf = create_alg(**_fhadp_kwargs(algorithm="INFADP", policy_func_name="DetermPolicy", value_func_type="MLP",
                                     value_func_name="StateValue", value_hidden_sizes=[64, 64],
                                     value_hidden_activation="gelu", value_learning_rate=1e-3))
    keys = list(inf.state_dict())
    assert keys[0] == "v.v.0.weight" and "v_target.v.4.bias" in keys and "policy_target.pi.0.weight" in keys
    # lr scheduler wiring like the reference's example scripts
    sch = create_alg(**_fhadp_kwargs(policy_scheduler={"name": "LinearLR", "params": {
        "start_factor": 1.0, "end_factor": 0.0, "total_iters": 10}}))
    assert "policy_scheduler" in sch.networks.scheduler_dict


def test_reference_random_init_is_reproduced():
    """Same seed -> same nn.Linear draws as the reference's FHADP constructor (fixture checksum)."""
    from conftest import load_golden
    from gops_amd.create_pkg.create_alg import create_alg
    g = load_golden("big_cfg1_idp_fhadp_b64_h10")
    torch.manual_seed(0)
    alg = create_alg(**_fhadp_kwargs())
    assert abs(alg.networks.policy.pi[0].weight.double().sum().item() - float(g["chk/policy_w0_sum"])) < 1e-9


def test_no_cpu_fallback():
    from gops_amd.create_pkg.create_alg import create_alg
    from gops_amd.utils.synthetic import make_batch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    alg = create_alg(**_fhadp_kwargs())
    with pytest.raises(RuntimeError, match="no CPU path"):
        alg.local_update(make_batch(dict(env_id="pyth_idpendulum", batch=8), 0), 0)


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from gops_amd.trainer.grad_sync import GradAllReducer, broadcast_parameters
from gops_amd.trainer.on_sync_trainer import OnSyncTrainer
dist.init_process_group("gloo")
r, n = dist.get_rank(), dist.get_world_size()

class FakeAlg:   # same update API as FHADP, CPU arithmetic: grad = mean over the local batch of x
    def __init__(self):
        torch.manual_seed(100 + r)   # deliberately different init per rank: trainer must broadcast
        self.networks = torch.nn.Linear(3, 1, bias=False)
        self.tb_info = {}
    def get_remote_update_info(self, data, it):
        return self.tb_info, {"grad": [data["obs"].mean(0, keepdim=True).clone()]}
    def remote_update(self, info):
        with torch.no_grad():
            self.networks.weight -= 0.5 * info["grad"][0]

class Sampler:
    networks = None
    def sample_with_replay_format(self):
        g = torch.Generator().manual_seed(r)
        return {"obs": torch.rand(4, 3, generator=g)}, {}
    def get_total_sample_number(self): return 0

alg = FakeAlg()
tr = OnSyncTrainer(alg, Sampler(), None, max_iteration=1, log_save_interval=10, apprfunc_save_interval=10,
                   eval_interval=10, save_folder=None, ini_network_dir=None, use_gpu=False)
w0 = alg.networks.weight.detach().clone()
gathered = [torch.zeros_like(w0) for _ in range(n)]
dist.all_gather(gathered, w0)
assert all(torch.equal(g, gathered[0]) for g in gathered), "replicas not broadcast from rank 0"
tr.step()
batches = [torch.rand(4, 3, generator=torch.Generator().manual_seed(k)) for k in range(n)]
expect = w0 - 0.5 * torch.cat(batches).mean(0, keepdim=True)   # gradient of the CONCATENATED batch
assert torch.allclose(alg.networks.weight, expect, atol=1e-7), (alg.networks.weight, expect)
red = GradAllReducer()
info = {"policy": [torch.full((5,), float(r)), torch.full((2, 2), 10.0 * r)], "v": [torch.ones(3) * (r + 1)], "iteration": 7}
red.average_(info)
assert info["iteration"] == 7   # MPG's update_info carries its iteration counter: not a gradient
assert torch.allclose(info["policy"][0], torch.full((5,), (n - 1) / 2))
assert torch.allclose(info["v"][0], torch.ones(3) * (n + 1) / 2)
# a network's gradients as allocated by the algorithms: views into one flat buffer, reduced in place
from gops_amd.algorithm.base import grad_buffers
from gops_amd.apprfunc.mlp import StateValue
from gops_amd.trainer.grad_sync import _as_one_buffer
net = StateValue(obs_dim=3, hidden_sizes=[16, 16], hidden_activation="relu", output_activation="linear",
                 action_distribution_cls=None)
gw, gb = grad_buffers(net)
grads = [p.grad for p in net.parameters()]
flat = _as_one_buffer(grads)
assert flat is not None and flat.data_ptr() == net._flat_grad.data_ptr() and flat.numel() == net._flat_grad.numel()
for i, g in enumerate(grads):
    g.fill_(float(r * 10 + i))
ptrs = [g.data_ptr() for g in grads]
red.average_({"v": grads})
assert [g.data_ptr() for g in grads] == ptrs
for i, g in enumerate(grads):
    assert torch.allclose(g, torch.full_like(g, 10 * (n - 1) / 2 + i))
assert _as_one_buffer([grads[0], grads[2]]) is None   # a gap between the pieces: not one buffer
# ---- off-policy variant: every rank owns a sampler + replay buffer, gradients of the replay batches are averaged ----
from gops_amd.trainer.off_sync_trainer import OffSyncTrainer
from gops_amd.trainer.buffer.replay_buffer import ReplayBuffer
import numpy as np

class OffSampler:
    networks = None
    def sample(self):
        g = np.random.RandomState(50 + r)
        return [(g.rand(3).astype(np.float32), np.zeros(1, np.float32), 0.0, False, {}, g.rand(3).astype(np.float32), {}, 0.0)
                for _ in range(8)], {}
    def get_total_sample_number(self): return 0

class ScaledAlg(FakeAlg):   # honours the deferred 1/N like the HIP Adam kernel does
    accepts_grad_scale = True
    def remote_update(self, info):
        with torch.no_grad():
            self.networks.weight -= 0.5 * info["grad"][0] * info.get("_grad_scale", 1.0)

alg2 = ScaledAlg()
buf = ReplayBuffer(index=r, trainer="off_sync_trainer", seed=9, obsv_dim=3, action_dim=1, buffer_max_size=64,
                   additional_info={}, buffer_device="cpu")
tr2 = OffSyncTrainer(alg2, OffSampler(), buf, None, max_iteration=1, log_save_interval=10, apprfunc_save_interval=10,
                     eval_interval=10, save_folder=None, ini_network_dir=None, use_gpu=False, buffer_warm_size=16,
                     replay_batch_size=8, sample_interval=1)
w0 = alg2.networks.weight.detach().clone()
gathered = [torch.zeros_like(w0) for _ in range(n)]
dist.all_gather(gathered, w0)
assert all(torch.equal(g, gathered[0]) for g in gathered), "off_sync replicas not broadcast from rank 0"
seen = {}
orig = alg2.get_remote_update_info
def spy(data, it):
    seen["mean"] = data["obs"].mean(0, keepdim=True).clone()
    return orig(data, it)
alg2.get_remote_update_info = spy
tr2.step()
means = [torch.zeros_like(seen["mean"]) for _ in range(n)]
dist.all_gather(means, seen["mean"])
expect = w0 - 0.5 * torch.stack(means).mean(0)      # mean over ranks of the per-rank replay-batch gradients
assert torch.allclose(alg2.networks.weight, expect, atol=1e-7), (alg2.networks.weight, expect)
dist.destroy_process_group()
open(os.path.join(sys.argv[2], f"ok_{r}"), "w").write("ok")
"""


def test_sync_trainer_gradient_exchange_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29671", str(script), ROOT, str(tmp_path)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()


@pytest.mark.skipif(not os.path.isdir("/root/reference/gops"), reason="needs the GOPS tree (build container only)")
def test_overlay_routes_hot_path_modules_only(tmp_path):
    """`gops.<hot path>` -> gops_amd, the rest of GOPS untouched (run in a subprocess: it edits sys.modules)."""
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import _ref_import; _ref_import.install()            # gym / tensorboard stubs + reference on sys.path
import gops_amd.overlay as ov; ov.install()
from gops.create_pkg.create_alg import create_alg
from gops.create_pkg.create_env_model import create_env_model
import gops.algorithm.fhadp as f, gops.apprfunc.mlp as m
assert create_alg.__module__ == "gops_amd.create_pkg.create_alg"
assert f.FHADP.__module__ == "gops_amd.algorithm.fhadp" and m.StateValue.__module__ == "gops_amd.apprfunc.mlp"
import gops.trainer.buffer.replay_buffer as rb, gops.utils.common_utils as cu, gops.trainer.sampler.off_sampler as smp
assert rb.ReplayBuffer.__module__ == "gops_amd.trainer.buffer.replay_buffer"      # device-resident buffer
assert cu.__file__.startswith("/root/reference") and smp.__file__.startswith("/root/reference")   # the reference's own
import numpy as np
buf = rb.ReplayBuffer(index=0, obsv_dim=6, action_dim=1, buffer_max_size=64, seed=0, additional_info={},
                      trainer="off_serial_trainer", buffer_device="cpu")
print("overlay ok")
""" % (ROOT, os.path.join(ROOT, "tests", "golden"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "overlay ok" in out.stdout, out.stdout + out.stderr


def _buffer_kwargs(**over):
    kw = dict(trainer="off_serial_trainer", seed=3, obsv_dim=5, action_dim=2, buffer_max_size=10,
              buffer_name="replay_buffer", buffer_device="cpu",
              additional_info={"state": {"shape": (3,), "dtype": np.float32}, "ref_time": {"shape": (), "dtype": np.float32}})
    kw.update(over)
    return kw


def _transition(i):
    info = {"state": np.full(3, i, np.float32), "ref_time": np.float32(i)}
    nxt = {"state": np.full(3, i + 0.5, np.float32), "ref_time": np.float32(i + 0.5)}
    return (np.full(5, i, np.float32), np.full(2, -i, np.float32), float(i), bool(i % 2), info,
            np.full(5, i + 1, np.float32), nxt, np.float32(0.1 * i))


def test_replay_buffer_interface_and_ring_semantics():
    """Same keys / dtypes / ring behaviour as the reference buffer (replay_buffer.py:27-108)."""
    from gops_amd.create_pkg.create_buffer import create_buffer, registry
    assert "replay_buffer" in registry
    assert create_buffer(**_buffer_kwargs(trainer="on_serial_trainer")) is None
    with pytest.raises(KeyError, match="No registered buffer with id: nope"):
        create_buffer(**_buffer_kwargs(buffer_name="nope"))
    buf = create_buffer(**_buffer_kwargs())
    buf.add_batch([_transition(i) for i in range(7)])
    assert len(buf) == 7 and buf.ptr == 7
    buf.store(*_transition(7))
    buf.add_batch([_transition(i) for i in range(8, 13)])        # wraps: rows 0..2 now hold 10, 11, 12
    assert len(buf) == 10 and buf.ptr == 3
    assert buf.buf["rew"].tolist() == [10, 11, 12, 3, 4, 5, 6, 7, 8, 9]
    batch = buf.sample_batch(64)
    assert set(batch) == {"obs", "obs2", "act", "rew", "done", "logp", "state", "next_state", "ref_time", "next_ref_time"}
    assert all(v.dtype == torch.float32 and v.shape[0] == 64 for v in batch.values())
    assert batch["state"].shape == (64, 3) and batch["ref_time"].shape == (64,)
    # rows stay consistent across keys, and every live row can be drawn
    assert torch.equal(batch["obs"][:, 0], batch["rew"]) and torch.equal(batch["next_state"][:, 0], batch["rew"] + 0.5)
    assert torch.equal(batch["done"], (batch["rew"] % 2 == 1).float())
    assert set(batch["rew"].tolist()) <= set(range(3, 13)) and len(set(batch["rew"].tolist())) >= 8
    again = create_buffer(**_buffer_kwargs())
    again.add_batch([_transition(i) for i in range(8)])
    again.add_batch([_transition(i) for i in range(8, 13)])
    assert torch.equal(again.sample_batch(64)["rew"], batch["rew"])   # seeded index stream
    again.add_tensors({"obs": torch.ones(2, 5), "rew": torch.tensor([100.0, 101.0])})
    assert again.buf["rew"][3:5].tolist() == [100.0, 101.0] and again.buf["obs"][3].tolist() == [1.0] * 5
    assert buf.__get_RAM__() > 0
    # a zero-size info slot (pyth_mobilerobot.py:88-92 declares "constraint" with shape (0,)): the reference's numpy buffer
    # broadcasts the [1] value into nothing and samples a [B, 0] tensor
    zero = create_buffer(**_buffer_kwargs(additional_info={"constraint": {"shape": (0,), "dtype": np.float32}}))
    t = _transition(1)
    zero.add_batch([t[:4] + ({"constraint": np.array([0.3])},) + t[5:6] + ({"constraint": np.array([0.1])},) + t[7:]] * 3)
    assert len(zero) == 3 and zero.sample_batch(4)["constraint"].shape == (4, 0)


def test_off_serial_trainer_loop(tmp_path):
    """warm-up, sample_interval, replay batch into alg.local_update, checkpoints - reference
    off_serial_trainer.py:30-165 semantics with a CPU stand-in algorithm."""
    from gops_amd.create_pkg.create_buffer import create_buffer
    from gops_amd.create_pkg.create_trainer import create_trainer

    class Alg:
        def __init__(self):
            self.networks = torch.nn.Linear(5, 2)
            self.seen = []

        def local_update(self, data, it):
            self.seen.append((it, data["obs"].shape[0], float(data["rew"].max())))
            return {"Loss/Actor loss-RL iter": 0.0}

    class Sampler:
        networks = None
        calls = 0

        def sample(self):
            base = 4 * Sampler.calls
            Sampler.calls += 1
            return [_transition(base + i) for i in range(4)], {"Time/Sampler time [ms]-RL iter": 1.0}

        def get_total_sample_number(self):
            return 4 * Sampler.calls

    alg, buf = Alg(), create_buffer(**_buffer_kwargs(buffer_max_size=100))
    tr = create_trainer(alg, Sampler(), buf, None, trainer="off_serial_trainer", buffer_name="replay_buffer",
                        replay_batch_size=16, buffer_warm_size=10, sample_interval=2, max_iteration=5,
                        log_save_interval=100, apprfunc_save_interval=4, eval_interval=100,
                        save_folder=str(tmp_path), ini_network_dir=None, use_gpu=False)
    assert len(buf) == 12 and Sampler.calls == 3             # warm-up: 3 sampler calls reach >= 10
    tr.train()
    assert Sampler.calls == 3 + 3                            # iterations 0, 2, 4 sample
    assert [s[:2] for s in alg.seen] == [(i, 16) for i in range(5)]
    assert len(buf) == 24
    assert {"apprfunc_0.pkl", "apprfunc_4.pkl", "apprfunc_5.pkl"} <= set(os.listdir(tmp_path / "apprfunc"))


def test_off_serial_trainer_with_prioritized_replay(tmp_path):
    """the per_flag path of the reference's off_serial_trainer.py:96-100 - with `buffer_name=
    "prioritized_replay_buffer"` the algorithm's `local_update` returns (tb_info, tree indices, new priorities) and the trainer
    hands them to `buffer.update_batch`; the trees must move, and sampling must follow the new priorities."""
    from gops_amd.create_pkg.create_buffer import create_buffer
    from gops_amd.create_pkg.create_trainer import create_trainer
    from gops_amd.trainer.buffer.prioritized_replay_buffer import PrioritizedReplayBuffer

    class Alg:
        def __init__(self):
            self.networks = torch.nn.Linear(5, 2)
            self.updates = []

        def local_update(self, data, it):
            # a TD-error-like priority: large for the transitions whose reward tag is a multiple of 5, tiny for the rest
            pr = torch.where(data["rew"].round() % 5 == 0, torch.full_like(data["rew"], 50.0), torch.full_like(data["rew"], 1e-3))
            self.updates.append((data["idx"].clone(), pr.clone()))
            return {"Loss/Actor loss-RL iter": 0.0}, data["idx"], pr

    class Sampler:
        networks = None
        calls = 0

        def sample(self):
            base = 4 * Sampler.calls
            Sampler.calls += 1
            return [_transition(base + i) for i in range(4)], {"Time/Sampler time [ms]-RL iter": 1.0}

        def get_total_sample_number(self):
            return 4 * Sampler.calls

    kw = _buffer_kwargs(buffer_max_size=64)
    kw["buffer_name"] = "prioritized_replay_buffer"
    buf = create_buffer(**kw)
    assert isinstance(buf, PrioritizedReplayBuffer)
    alg = Alg()
    tr = create_trainer(alg, Sampler(), buf, None, trainer="off_serial_trainer", buffer_name="prioritized_replay_buffer",
                        replay_batch_size=16, buffer_warm_size=40, sample_interval=100, max_iteration=6,
                        log_save_interval=100, apprfunc_save_interval=100, eval_interval=100,
                        save_folder=str(tmp_path), ini_network_dir=None, use_gpu=False)
    assert tr.per_flag
    sum0 = buf.sum_tree[0].item()
    tr.train()
    assert len(alg.updates) == 6
    # the trees changed, and stay consistent: every inner node is the sum / min of its children
    assert buf.sum_tree[0].item() != sum0
    n = buf.max_size
    inner = torch.arange(n - 1)
    assert torch.allclose(buf.sum_tree[inner], buf.sum_tree[2 * inner + 1] + buf.sum_tree[2 * inner + 2], rtol=1e-12, atol=0)
    assert torch.equal(buf.min_tree[inner], torch.minimum(buf.min_tree[2 * inner + 1], buf.min_tree[2 * inner + 2]))
    # the leaves the algorithm marked carry (|p| + eps)^alpha of ITS priorities, and sampling now prefers them
    idx, pr = alg.updates[-1]
    want = (pr.double() + buf.epsilon) ** buf.alpha
    assert torch.allclose(buf.sum_tree[idx.long()], want, rtol=1e-12)
    batch = buf.sample_batch(64)
    assert (batch["rew"].round() % 5 == 0).float().mean() > 0.5   # 1 in 5 stored transitions, most of the sampled ones


class _SyncTrap:
    """Stands for a device scalar: any attempt to read it on the host (a stream sync on a GPU) raises."""

    def __neg__(self):
        return self

    def __getitem__(self, i):
        return self

    def dim(self):   # shape queries are host-side metadata, not a read of the value
        return 0

    def _boom(self, *a, **k):
        raise AssertionError("host read of a device scalar between the backward pass and the all-reduce")

    item = tolist = __float__ = __int__ = cpu = _boom


def test_remote_update_path_has_no_host_sync_before_the_collective(monkeypatch):
    """`get_remote_update_info` must queue the gradient kernels and return WITHOUT reading the
    loss back (`.item()` blocks the host until the backward sweep has drained, serialising the all-reduce
    behind it); the scalars stay lazy in tb_info and the 1/N goes to the Adam kernel as `_grad_scale`."""
    from gops_amd.create_pkg.create_alg import create_alg
    from gops_amd.trainer.grad_sync import GradAllReducer
    from gops_amd.utils.tensorboard_setup import tb_tags
    # (precision_check_interval=0: the PrecisionGuard's check - one deliberate host read every 500 gradients - is not what this
    #  test is about; tests/test_trained256_gpu.py covers it)
    alg = create_alg(**_fhadp_kwargs(precision_check_interval=0))
    monkeypatch.setattr(alg, "_device_batch", lambda d: d)
    monkeypatch.setattr(alg._grad_graph, "run", lambda *a, **k: _SyncTrap())
    monkeypatch.setattr(alg, "_log", lambda *a, **k: (_ for _ in ()).throw(AssertionError("_log syncs the host")))
    tb, info = alg.get_remote_update_info({"obs": torch.zeros(4, 6), "done": torch.zeros(4)}, 0)
    assert isinstance(tb[tb_tags["loss_actor"]], _SyncTrap) and list(info) == ["grad"]
    with pytest.raises(AssertionError):
        float(tb[tb_tags["loss_actor"]])
    # single process: the reducer leaves everything alone; N ranks would add "_grad_scale" = 1/N
    assert GradAllReducer().average_(info, defer_scale=True) is info and "_grad_scale" not in info
    assert getattr(alg, "accepts_grad_scale", False)

    kw = _fhadp_kwargs()
    kw.update(algorithm="INFADP", policy_func_name="DetermPolicy", value_func_type="MLP", value_func_name="StateValue",
              value_hidden_sizes=[64, 64], value_hidden_activation="gelu", value_learning_rate=1e-3)
    kw.pop("pre_horizon", None)
    inf = create_alg(**kw, precision_check_interval=0)
    import gops_amd.algorithm.infadp as infadp_mod
    monkeypatch.setattr(infadp_mod, "batch_to_device", lambda d, dev, keys: d)
    monkeypatch.setattr(infadp_mod, "cuda_device_of", lambda nets: torch.device("cpu"))
    monkeypatch.setattr(inf, "_gradient_kernels", lambda mode, b: _SyncTrap())
    monkeypatch.setattr(inf, "_log", lambda *a, **k: (_ for _ in ()).throw(AssertionError("_log syncs the host")))
    for it, name in ((0, "v"), (1, "policy")):
        tb, info = inf.get_remote_update_info({"obs": torch.zeros(4, 6)}, it)
        assert list(info) == [name]


def test_networks_with_cached_abi_views_deepcopy():
    """the ctypes `GopsMlp` views (raw device pointers) used to live in the modules' __dict__ and
    made `copy.deepcopy(networks)` - what the trainers do for host-side samplers - raise.  They now live in a weak
    side table keyed by module."""
    import copy
    from gops_amd import hip_backend as hb
    from gops_amd.apprfunc import mlp as mlp_mod
    from gops_amd.create_pkg.create_alg import create_alg
    alg = create_alg(**_fhadp_kwargs())
    pol = alg.networks.policy
    view = hb.GopsMlp()
    view.weight[0] = 0x1000   # a struct with pointer fields: not picklable, not deep-copyable
    mlp_mod._HIP_CACHE[pol] = (("stale",), view)
    with pytest.raises(ValueError):
        copy.deepcopy(view)
    twin = copy.deepcopy(alg.networks)
    assert twin.policy is not pol and twin.policy not in mlp_mod._HIP_CACHE
    assert all(torch.equal(a, b) for a, b in zip(twin.state_dict().values(), alg.networks.state_dict().values()))
    assert not any(k.startswith("_hip") for k in vars(pol))


def test_trainer_calls_ray_actor_evaluators_through_remote(monkeypatch, tmp_path):
    """GOPS's create_evaluator returns a Ray actor handle; its methods are `.remote()` calls."""
    import types
    from gops_amd.trainer import _common

    class _Method:
        def __init__(self, fn):
            self.fn = fn

        def __call__(self, *a):
            raise TypeError("Actor methods cannot be called directly")

        def remote(self, *a):
            return ("ref", self.fn(*a))

    class _Actor:
        loaded = None

        def __init__(self):
            self.load_state_dict = _Method(lambda sd: setattr(_Actor, "loaded", sorted(sd)))
            self.run_evaluation = _Method(lambda it: 12.5 + it)

    monkeypatch.setitem(sys.modules, "ray", types.SimpleNamespace(get=lambda ref: ref[1]))
    assert _common.call_maybe_remote(_Actor(), "run_evaluation", 3) == 15.5

    class _Plain:
        def run_evaluation(self, it):
            return float(it)
    assert _common.call_maybe_remote(_Plain(), "run_evaluation", 4) == 4.0


_PLUMBING = r"""
import os, sys, types, runpy
ROOT, SCRIPT, SAVE = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import _ref_import; _ref_import.install()            # gym / tensorboard stubs + /root/reference on sys.path
import numpy as np
for _old, _new in (("float_", np.float64), ("int_", np.int64), ("bool8", np.bool_)):   # the reference targets numpy 1.x
    if not hasattr(np, _old):
        setattr(np, _old, _new)

# ---- a minimal in-process `ray`: actor handles whose methods are `.remote()` calls --------------------
class _Method:
    def __init__(self, fn): self._fn = fn
    def __call__(self, *a, **k): raise TypeError("Actor methods cannot be called directly")
    def remote(self, *a, **k): return self._fn(*a, **k)
class _Handle:
    def __init__(self, obj): object.__setattr__(self, "_obj", obj)
    def __getattr__(self, name): return _Method(getattr(self._obj, name))
def _remote(*dargs, **dkw):
    def wrap(cls):
        return types.SimpleNamespace(remote=lambda *a, **k: _Handle(cls(*a, **k)))
    return wrap(dargs[0]) if len(dargs) == 1 and callable(dargs[0]) and not dkw else wrap
sys.modules["ray"] = types.SimpleNamespace(init=lambda *a, **k: None, remote=_remote, get=lambda x: x, put=lambda x: x,
                                           shutdown=lambda: None, is_initialized=lambda: True)

import gops_amd.overlay as ov; ov.install()           # gops.<hot path> -> gops_amd
import gops.utils.plot_evaluation as pe; pe.plot_all = lambda *a, **k: None
import gops.utils.tensorboard_setup as ts             # no tensorboard package in this image: post-processing helpers off
ts.start_tensorboard = ts.save_tb_to_csv = lambda *a, **k: None
import gops_amd.trainer._common as tc
seen = {}
def train_one(self):                                   # the script calls trainer.train(): one step is enough here
    seen["trainer"] = type(self).__module__ + "." + type(self).__name__
    seen["alg"] = type(self.alg).__module__
    seen["buffer"] = type(self.buffer).__module__ if getattr(self, "buffer", None) is not None else None
    seen["sampler"] = type(getattr(self.sampler, "_obj", self.sampler)).__module__   # (async scripts: an actor handle)
    seen["evaluator"] = type(self.evaluator).__name__
    seen["warm"] = len(self.buffer) if getattr(self, "buffer", None) is not None else None
    try:
        self.step()
        seen["step"] = "ran"
    except RuntimeError as e:
        seen["step"] = str(e)
tc.TrainerBase.train = train_one
import gops_amd.trainer.off_async_trainer as oat
oat.OffAsyncTrainer.train = train_one
sys.argv = [SCRIPT, "--save_folder", SAVE, "--max_iteration", "2", "--buffer_warm_size", "128", "--sample_batch_size", "64"]
runpy.run_path(SCRIPT, run_name="__main__")
import torch
want = "off_async_trainer.OffAsyncTrainer" if "_async" in os.path.basename(SCRIPT) else "off_serial_trainer.OffSerialTrainer"
assert seen["trainer"] == "gops_amd.trainer." + want, seen
assert seen["alg"].startswith("gops_amd.algorithm."), seen
assert seen["buffer"] == "gops_amd.trainer.buffer.replay_buffer", seen
assert seen["sampler"].startswith("gops.trainer.sampler"), seen          # the reference's own numpy-env sampler
assert seen["evaluator"] == "_Handle" and seen["warm"] >= 128, seen
if not torch.cuda.is_available():
    assert "no CPU path" in seen["step"], seen                              # the ONLY thing missing without a GPU
else:
    assert seen["step"] == "ran", seen
print("plumbing ok", seen["alg"])
"""


@pytest.mark.skipif(not os.path.isdir("/root/reference/gops"), reason="needs the GOPS tree (build container only)")
@pytest.mark.parametrize("script", ["example_train/fhadp/fhadp_mlp_idpendulum_serial.py",
                                    "example_train/fhadp/fhadp_mlp_veh3dofconti_serial.py",
                                    "example_train/infadp/infadp_mlp_lqs4a2_offserial.py",
                                    "example_train/fhadp/fhadp_mlp_veh3dofconti_surrcstr_penalty_serial.py",
                                    "example_train/fhadp/fhadp_mlp_lqs3a1_serial.py",
                                    "example_train/fhadp/fhadp_mlp_veh2dofconti_serial.py",
                                    "example_train/infadp/infadp_mlp_veh2dofconti_offserial.py",
                                    "example_train/spil/spil_mlp_veh3dofconti_errcstr_offserial.py",
                                    "example_train/spil/spil_mlp_veh3dofconti_surrcstr_offserial.py",
                                    "example_train/spil/spil_mlp_veh2dofconti_errcstr_offserial.py",
                                    "example_train/spil/spil_mlp_mobilerobot_offserial.py",
                                    "example_train/mpg/mpg_mlp_cartpoleconti_offserial.py",
                                    "example_train/fhadp/fhadp2_mlp_veh3dofconti_serial.py",
                                    "example_train/fhadp/fhadp_mlp_lqs2a1_serial.py",
                                    "example_train/fhadp/fhadp_mlp_lqs5a1_serial.py",
                                    "example_train/infadp/infadp_mlp_idpendulum_serial.py",
                                    "example_train/infadp/infadp_mlp_cartpoleconti_offserial.py",
                                    "example_train/infadp/infadp_mlp_veh3dofconti_offserial.py",
                                    "example_train/infadp/infadp_mlp_lqs6a3_offserial.py",
                                    "example_train/mac/mac_mlp_cartpoleconti_offserial.py",
                                    # off_async scripts: `create_alg` hands out a list of actor-like handles
                                    # (`alg_id.set_parameters.remote(...)`), the samplers are actor handles of the reference
                                    "example_train/fhadp/fhadp_mlp_idpendulum_async.py",
                                    "example_train/infadp/infadp_mlp_cartpoleconti_async.py",
                                    "example_train/mpg/mpg_mlp_cartpoleconti_async.py",
                                    "example_train/spil/spil_mlp_mobilerobot_async.py",
                                    "example_train/mac/mac_mlp_cartpoleconti_async.py"])   # (the pendulum scripts need the gym package for their data env)
def test_example_scripts_run_unchanged_through_the_overlay(script, tmp_path):
    """BASELINE configs[0] plumbing: the reference's UNMODIFIED example scripts (their own argparse block, create_env,
    init_args, create_sampler, create_evaluator - a Ray actor handle, here from an in-process stub) executed with
    `gops_amd.overlay` installed: create_alg / create_buffer / create_trainer resolve to this package, the trainer
    warms its buffer from the reference's numpy-env sampler, and the first update fails only for want of a GPU."""
    worker = tmp_path / "plumbing.py"
    worker.write_text(_PLUMBING)
    out = subprocess.run([sys.executable, str(worker), ROOT, os.path.join("/root/reference", script), str(tmp_path / "run")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "plumbing ok" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


@pytest.mark.parametrize("name", ["dataenv_veh_p10", "dataenv_lq_s4a2"])
def test_replay_buffer_matches_reference_buffer_side_by_side(name):
    """The reference's ReplayBuffer (gops/trainer/buffer/replay_buffer.py) was fed 150 recorded transitions at capacity
    100 by tests/golden/make_golden.py (ring wrap included) and its arrays stored in the fixture; this package's buffer,
    fed the same transitions through `add_batch` (sampler tuples) and through `add_tensors` (batched device insert),
    must hold exactly the same rows, pointer and size, and sample float32 rows of them."""
    import json
    from gops_amd.trainer.buffer.replay_buffer import ReplayBuffer
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    g = {k: z[k] for k in z.files}
    kw = json.loads(str(g["meta/buffer_kwargs"]))
    kw["additional_info"] = {k: {"shape": tuple(v["shape"]), "dtype": np.float32} for k, v in kw["additional_info"].items()}
    t = {k[2:]: g[k] for k in g if k.startswith("t/")}
    info_keys = list(kw["additional_info"])
    rows = []
    for i in range(150):
        info = {k: t["info_" + k][i] for k in info_keys}
        nxt = {k: t["next_" + k][i] for k in info_keys}
        rows.append((t["obs"][i], t["act"][i], float(t["rew"][i]), bool(t["done"][i]), info, t["obs2"][i], nxt, 0.25))
    a = ReplayBuffer(index=0, buffer_device="cpu", **kw)
    for lo in range(0, 150, 7):            # sampler-sized chunks
        a.add_batch(rows[lo:lo + 7])
    b = ReplayBuffer(index=0, buffer_device="cpu", **kw)
    for lo in range(0, 150, 32):           # batched tensor insert (DeviceEnvSampler path)
        sl = slice(lo, min(lo + 32, 150))
        batch = dict(obs=torch.from_numpy(t["obs"][sl]), act=torch.from_numpy(t["act"][sl]), rew=torch.from_numpy(t["rew"][sl]),
                     done=torch.from_numpy(t["done"][sl]), obs2=torch.from_numpy(t["obs2"][sl]),
                     logp=torch.full((sl.stop - sl.start,), 0.25))
        for k in info_keys:
            batch[k], batch["next_" + k] = torch.from_numpy(t["info_" + k][sl]), torch.from_numpy(t["next_" + k][sl])
        b.add_tensors(batch)
    for buf in (a, b):
        assert buf.size == int(g["buf/size"]) and buf.ptr == int(g["buf/ptr"]) and len(buf) == 100
        assert sorted(buf.buf) == sorted(k[len("buf/store/"):] for k in g if k.startswith("buf/store/"))
        for k, v in buf.buf.items():
            assert np.array_equal(v.numpy(), g["buf/store/" + k].astype(np.float32)), k
    s = a.sample_batch(16)
    assert set(s) == set(a.buf) and all(v.dtype == torch.float32 and v.shape[0] == 16 for v in s.values())
    stored = {tuple(r.tolist()) for r in a.buf["obs"]}
    assert all(tuple(r.tolist()) in stored for r in s["obs"])


_ASYNC_WORKER = r"""
import os, sys, time, torch, torch.distributed as dist, numpy as np
sys.path.insert(0, sys.argv[1])
from gops_amd.trainer.off_async_trainer import OffAsyncTrainer
from gops_amd.trainer.buffer.replay_buffer import ReplayBuffer
dist.init_process_group("gloo")
r, n = dist.get_rank(), dist.get_world_size()

class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(7 + r)                         # different init per rank: the trainer must broadcast rank 0's
        self.policy = torch.nn.Linear(3, 2)

class Alg:   # FHADP-shaped update API; the gradient of rank r is the constant r + 1 (so the center's weights tell who was applied)
    accepts_grad_scale = True
    def __init__(self):
        self.networks = Net()
        self.tb_info = {}
        self.seen_weights = []
    def get_remote_update_info(self, data, it):
        self.seen_weights.append(self.networks.policy.weight.detach().clone())
        time.sleep(0.015 if r == 1 else 0.005)           # gradients take time; a slow worker must not hold the others back
        return self.tb_info, {"grad": [torch.full_like(p, float(r + 1)) for p in self.networks.policy.parameters()]}
    def remote_update(self, info):
        with torch.no_grad():
            for p, g in zip(self.networks.policy.parameters(), info["grad"]):
                p -= 0.125 * g

class Sampler:
    networks = None
    def sample(self):
        g = np.random.RandomState(r)
        return [(g.rand(3).astype(np.float32), np.zeros(1, np.float32), 0.0, False, {}, g.rand(3).astype(np.float32), {}, 0.0)
                for _ in range(8)], {}
    def get_total_sample_number(self): return 0

alg = Alg()
buf = ReplayBuffer(index=r, trainer="off_async_trainer", seed=1, obsv_dim=3, action_dim=1, buffer_max_size=64,
                   additional_info={}, buffer_device="cpu")
tr = OffAsyncTrainer(alg, Sampler(), buf, None, max_iteration=40, log_save_interval=1000, apprfunc_save_interval=1000,
                     eval_interval=10 ** 9, save_folder=None, ini_network_dir=None, use_gpu=False, buffer_warm_size=16,
                     replay_batch_size=8, sample_interval=2)
w0 = alg.networks.policy.weight.detach().clone()
tr.train()
final = alg.networks.policy.weight.detach().clone()
if r == 0:
    assert tr.iteration == 40 and sum(tr.applied_from) == 40, (tr.iteration, tr.applied_from)
    assert all(c > 0 for c in tr.applied_from), tr.applied_from      # every rank contributed, nobody was starved
    expect = w0 - 0.125 * sum((k + 1) * c for k, c in enumerate(tr.applied_from))
    assert torch.allclose(final, expect, atol=1e-5), (final, expect, tr.applied_from)
else:
    # a worker computed each gradient on weights it RECEIVED from the center (stale by at most its own round trip)
    assert len(alg.seen_weights) >= 1 and tr._stop
gathered = [torch.zeros_like(final) for _ in range(n)]
dist.all_gather(gathered, final)
assert all(torch.equal(g, gathered[0]) for g in gathered), "workers must end on the center's final weights"
dist.destroy_process_group()
open(os.path.join(sys.argv[2], f"ok_{r}"), "w").write("ok")
"""


def test_off_async_trainer_applies_gradients_in_arrival_order_world_size_3(tmp_path):
    """Center network on rank 0 + two workers (gloo): 40 applied gradients, every rank contributes, the center's weights are
    exactly w0 - lr * sum of the applied per-rank gradients, and every rank ends on the center's final weights."""
    script = tmp_path / "worker.py"
    script.write_text(_ASYNC_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                          "--master-addr", "127.0.0.1", "--master-port", "29683", str(script), ROOT, str(tmp_path)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert all((tmp_path / f"ok_{k}").exists() for k in range(3))


_ASYNC_MULTINET_WORKER = r"""
import os, sys, time, torch, torch.distributed as dist, numpy as np
sys.path.insert(0, sys.argv[1])
from gops_amd.trainer.off_async_trainer import OffAsyncTrainer
from gops_amd.trainer.buffer.replay_buffer import ReplayBuffer
dist.init_process_group("gloo")
r, n = dist.get_rank(), dist.get_world_size()
STYLE = sys.argv[3]

class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.q1, self.q2, self.policy = torch.nn.Linear(3, 2), torch.nn.Linear(3, 1), torch.nn.Linear(3, 2)
        self.q1_target = torch.nn.Linear(3, 2)
        for p in self.q1_target.parameters():
            p.requires_grad = False
        self.net_dict = {"q1": self.q1, "q2": self.q2, "policy": self.policy}

class Alg:   # MPG-shaped ("<net>_grad" lists + the iteration counter) or SPIL-shaped ("v"-like subsets, here q1 / policy alternating + both)
    def __init__(self):
        self.networks = Net()
        self.tb_info = {}
        self.applied = []
    def get_remote_update_info(self, data, it):
        time.sleep(0.004)
        nets = self.networks
        g = lambda mod, c: [torch.full_like(p, c) for p in mod.parameters()]
        if STYLE == "mpg":
            return self.tb_info, {"q1_grad": g(nets.q1, 1.0 + r), "q2_grad": g(nets.q2, 10.0 + r), "policy_grad": g(nets.policy, 100.0 + r),
                                  "iteration": 1000 * r + it}
        names = [["q1"], ["policy"], ["q1", "policy"]][it % 3]
        return self.tb_info, {nm: g(nets.net_dict[nm], {"q1": 1.0, "policy": 100.0}[nm] + r) for nm in names}
    def remote_update(self, info):
        rec = {}
        for k, v in info.items():
            if isinstance(v, list):
                net = k[:-5] if k.endswith("_grad") else k
                assert len(v) == 2 and all(t.shape == p.shape for t, p in zip(v, self.networks.net_dict[net].parameters())), k
                assert all(torch.all(t == v[0].flatten()[0]) for t in v), k
                rec[k] = float(v[0].flatten()[0])
                with torch.no_grad():
                    for p_, g_ in zip(self.networks.net_dict[net].parameters(), v):
                        p_ -= 0.001 * g_
            else:
                rec[k] = v
        self.applied.append(rec)

class Sampler:
    networks = None
    def sample(self):
        g = np.random.RandomState(r)
        return [(g.rand(3).astype(np.float32), np.zeros(1, np.float32), 0.0, False, {}, g.rand(3).astype(np.float32), {}, 0.0)
                for _ in range(8)], {}
    def get_total_sample_number(self): return 0

alg = Alg()
buf = ReplayBuffer(index=r, trainer="off_async_trainer", seed=1, obsv_dim=3, action_dim=1, buffer_max_size=64,
                   additional_info={}, buffer_device="cpu")
tr = OffAsyncTrainer(alg, Sampler(), buf, None, max_iteration=30, log_save_interval=1000, apprfunc_save_interval=1000,
                     eval_interval=10 ** 9, save_folder=None, ini_network_dir=None, use_gpu=False, buffer_warm_size=16,
                     replay_batch_size=8, sample_interval=2)
assert [nm for nm, _, _ in tr._slots] == ["policy", "q1", "q2"]          # trainable nets only, sorted: the frozen target has no slot
tr.train()
if r == 0:
    assert len(alg.applied) == 30 and tr.applied_from[1] > 0, tr.applied_from
    remote = 0
    for rec in alg.applied:
        if STYLE == "mpg":   # every list and the scalar arrive, each from ONE source rank
            src = int(rec["q1_grad"] - 1.0)
            assert rec == {"q1_grad": 1.0 + src, "q2_grad": 10.0 + src, "policy_grad": 100.0 + src, "iteration": rec["iteration"]}, rec
            assert rec["iteration"] // 1000 == src
            remote += src
        else:                # one or two nets per message, never a dropped one
            assert set(rec) in ({"q1"}, {"policy"}, {"q1", "policy"}), rec
            srcs = {int(v - {"q1": 1.0, "policy": 100.0}[k]) for k, v in rec.items()}
            assert len(srcs) == 1
            remote += srcs.pop()
    assert remote == tr.applied_from[1]
    if STYLE != "mpg":
        assert any(len(rec) == 2 for rec in alg.applied)
dist.destroy_process_group()
open(os.path.join(sys.argv[2], f"ok_{r}"), "w").write("ok")
"""


@pytest.mark.parametrize("style,port", [("mpg", 29687), ("spil", 29688)])
def test_off_async_trainer_ships_every_gradient_list_and_scalar(tmp_path, style, port):
    """update_info of the multi-network algorithms crosses the wire whole: MPG's {q1_grad, q2_grad, policy_grad, iteration}
    and SPIL's per-iteration subsets ({v}, {policy} or both) - world size 2, gloo."""
    script = tmp_path / "worker.py"
    script.write_text(_ASYNC_MULTINET_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, str(tmp_path), style],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert all((tmp_path / f"ok_{k}").exists() for k in range(2))


def test_prioritized_replay_buffer_matches_reference_side_by_side():
    """tests/golden/per_buffer.npz: the reference's PrioritizedReplayBuffer (numpy sum / min trees walked in Python) run by
    make_golden.py on 50 synthetic transitions at capacity 37 - not a power of two, ring wrap included -, three sampled batches
    (unit draws recorded), three priority updates (one with a duplicated index).  This package's device buffer with vectorised
    tree operations, fed the same transitions / draws / priorities, must hold the same trees after every step and return the
    same leaves, weights and rows."""
    import json
    from gops_amd.trainer.buffer.prioritized_replay_buffer import PrioritizedReplayBuffer
    z = np.load(os.path.join(ROOT, "tests", "golden", "per_buffer.npz"))
    g = {k: z[k] for k in z.files}
    kw = json.loads(str(g["meta/buffer_kwargs"]))
    buf = PrioritizedReplayBuffer(index=0, buffer_device="cpu", **kw)
    t = {k[2:]: g[k] for k in g if k.startswith("t/")}

    def store(lo, hi, chunk):
        for a in range(lo, hi, chunk):
            b = min(a + chunk, hi)
            buf.add_batch([(t["obs"][i], t["act"][i], float(t["rew"][i]), bool(t["done"][i]), {}, t["obs2"][i], {}, 0.25) for i in range(a, b)])

    def check(tag):
        np.testing.assert_allclose(buf.sum_tree.numpy(), g[tag + "/sum_tree"], rtol=1e-13, atol=0)
        np.testing.assert_allclose(buf.min_tree.numpy(), g[tag + "/min_tree"], rtol=1e-13, atol=0)   # (torch's pow vs numpy's: 1 ulp)
        assert abs(float(buf.max_priority) - float(g[tag + "/max_priority"])) <= 1e-13 * float(g[tag + "/max_priority"])
        assert (buf.size, buf.ptr) == (int(g[tag + "/size"]), int(g[tag + "/ptr"]))

    store(0, 20, 7)
    check("s0")
    for k in range(3):
        if k == 1:
            store(20, 50, 11)
            check("s1")
        u = torch.from_numpy(g[f"b{k}/u"])
        rand = torch.rand
        torch.rand = lambda *a, **kw_: u.clone()     # the recorded unit draws instead of the device generator's
        try:
            b = buf.sample_batch(8)
        finally:
            torch.rand = rand
        np.testing.assert_array_equal(b["idx"].numpy(), g[f"b{k}/idx"])
        np.testing.assert_allclose(b["weight"].numpy(), g[f"b{k}/weight"], rtol=1e-6)
        np.testing.assert_array_equal(b["obs"].numpy(), g[f"b{k}/obs"])
        np.testing.assert_array_equal(b["rew"].numpy(), g[f"b{k}/rew"])
        assert b["idx"].dtype == torch.int32 and b["weight"].dtype == torch.float32
        buf.update_batch(torch.from_numpy(g[f"b{k}/upd_idx"]), torch.from_numpy(g[f"b{k}/upd_pr"]))
        check(f"u{k}")
        assert abs(buf.beta - float(g[f"u{k}/beta"])) < 1e-12
    from gops_amd.create_pkg.create_buffer import create_buffer
    assert isinstance(create_buffer(buffer_name="prioritized_replay_buffer", buffer_device="cpu", **kw), PrioritizedReplayBuffer)


_GUARD_WORKER = r"""
import os, sys, warnings
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from gops_amd.algorithm.base import PrecisionGuard
dist.init_process_group("gloo")
r = dist.get_rank()
# the SAME network on both replicas, but only rank 1's batch shows a distance beyond the threshold: the decision is taken with a
# MAX over the ranks, so BOTH move to the exact kernels (replicas that chose kernels on their own would stop computing the same update)
g = PrecisionGuard(interval=1, threshold=1e-4)
g.lockstep = True   # what on_sync / off_sync trainers declare (AlgorithmBase.set_lockstep_replicas); never inferred from dist.is_initialized()
base = torch.ones(8)
def flat_gradient(flags):
    exact = bool(flags & PrecisionGuard.exact_rollout_flags())
    return base.clone() if exact else base * (1.0 + (1e-3 if r == 1 else 1e-6))
assert g.due()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    d = g.check(flat_gradient)
assert g.exact and abs(d - 1e-3) < 1e-5, (r, g.exact, d)
assert any("exact-fp32 rollout kernels" in str(x.message) for x in w)
assert g.flags() & PrecisionGuard.exact_rollout_flags() == PrecisionGuard.exact_rollout_flags() and not g.due()
# ... and a distance below the threshold on every rank leaves both on the plane-split kernels
g2 = PrecisionGuard(interval=1, threshold=1e-4)
g2.lockstep = True
g2.due()
d2 = g2.check(lambda flags: base.clone() if flags & PrecisionGuard.exact_rollout_flags() else base * (1.0 + 1e-6 * (r + 1)))
assert not g2.exact and abs(d2 - 2e-6) < 1e-7, (r, d2)
# ranks that are NOT in lockstep (off_async_trainer: own gradient counts, point-to-point messages) decide locally - no collective
# is entered, so a rank that checks while the other does not cannot hang: only rank 1 checks here, and only rank 1 trips
g3 = PrecisionGuard(interval=1, threshold=1e-4)
assert not g3.lockstep
if r == 1:
    g3.due()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g3.check(flat_gradient)
assert g3.exact == (r == 1)
# a non-finite logged loss forces the next gradient to be a checked one (local decision only)
g4 = PrecisionGuard(interval=1000, threshold=1e-4)
assert g4.due() and not g4.due()
g4.observe_loss(float("nan"))
assert g4.due() and not g4.due()
g4.lockstep = True
g4.observe_loss(float("inf"))
assert not g4.due()
dist.barrier()
dist.destroy_process_group()
open(os.path.join(sys.argv[2], f"ok_{r}"), "w").write("ok")
"""


def test_precision_guard_decides_for_all_replicas_world_size_2(tmp_path):
    """Data-parallel replicas take the plane-split / exact-kernel decision TOGETHER (MAX all-reduce of the measured distance)."""
    script = tmp_path / "worker.py"
    script.write_text(_GUARD_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29677", str(script), ROOT, str(tmp_path)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()
