#!/usr/bin/env python
"""Benchmark of the GOPS ADP hot path on MI355X.

A "step" is one trainer iteration of the hot path on one batch: FHADP `compute_gradient` (fused
forward rollout + backward sweep + weight-gradient GEMMs through libgops_hip.so), the gradient
all-reduce when N > 1, and the Adam update.  Inputs are synthetic and already resident in HBM.
Metric (BASELINE.json): env-model steps/s = N * B * H * K / wall time, weak scaling (per-GPU batch
fixed).  N > 1:  python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...  - or plain
`python bench.py --gpus N`, which starts the N ranks itself.  The default 1-GPU run also times every other
BASELINE.json workload briefly and reports them in the `workloads` field of the one JSON line.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from gops_amd import hip_backend as hb  # noqa: E402
from gops_amd.create_pkg.create_alg import create_alg  # noqa: E402
from gops_amd.trainer.grad_sync import GradAllReducer  # noqa: E402
from gops_amd.utils.synthetic import CONFIGS, act_dim_of, make_batch, obs_dim_of  # noqa: E402

# MI355X peaks (MI355X_MICROARCH.md): dense fp32 matrix = fp32 vector rate; dense fp16/bf16 MFMA; HBM3E spec
MFMA_PEAK_TFLOPS = {"f32": 157.3, "f16": 2500.0}
HBM_PEAK_GBS = 8000.0
KERNEL_NAMES = {0: "rollout_fwd_kernel", 1: "rollout_bwd_kernel", 2: "dw_gemm_kernel(+reduce)",
                3: "value_fwd (rollout_fwd_kernel<ENV_NONE>)", 4: "value_bwd (rollout_bwd_kernel<ENV_NONE>)",
                5: "value_dw_gemm(+reduce)"}


def alg_kwargs(cfg, seed):
    A = act_dim_of(cfg)
    kw = dict(algorithm=cfg["alg"], trainer="on_sync_trainer", seed=seed, cnn_shared=False,
              env_id=cfg["env_id"], obsv_dim=obs_dim_of(cfg), action_dim=A, action_type="continu",
              action_high_limit=np.ones(A, dtype=np.float32), action_low_limit=-np.ones(A, dtype=np.float32),
              policy_func_type="MLP",
              policy_func_name="FiniteHorizonPolicy" if cfg["alg"] == "FHADP" else "DetermPolicy",
              policy_hidden_sizes=list(cfg["hidden"]), policy_hidden_activation=cfg["act"],
              policy_act_distribution="default", policy_learning_rate=1e-3, use_gpu=True)
    if cfg["alg"] == "FHADP":
        kw["pre_horizon"] = cfg.get("pre_horizon", cfg["horizon"])
        kw["gamma"] = cfg["gamma"]
    else:
        kw.update(value_func_type="MLP", value_func_name="StateValue", value_hidden_sizes=list(cfg["hidden"]),
                  value_hidden_activation=cfg["act"], value_learning_rate=1e-3)
        if "pre_horizon" in cfg:
            kw["pre_horizon"] = cfg["pre_horizon"]
    if "lq_config" in cfg:
        kw["lq_config"] = cfg["lq_config"]
    return kw


def mlp_sizes(cfg, net="policy"):
    if net == "value":
        return [obs_dim_of(cfg)] + list(cfg["hidden"]) + [1]
    return [obs_dim_of(cfg) + (1 if cfg["alg"] == "FHADP" else 0)] + list(cfg["hidden"]) + [act_dim_of(cfg)]


def mac_per_step(cfg, net="policy"):
    sizes = mlp_sizes(cfg, net)
    return sum(a * b for a, b in zip(sizes[:-1], sizes[1:]))


def bytes_per_step(cfg, dtype):
    """the activation stash written once and read once = 2 * sizeof * (in + sum(hidden) + A)."""
    sizes = mlp_sizes(cfg)
    return 2 * (2 if dtype == "f16" else 4) * sum(sizes)


PREHEAT_UPDATES = 150   # untimed updates in front of the warm-up steps (run_workload): the GPU's clock ramp takes ~40 updates / 20 ms at the
                        # target; 150 + W + K stays below the precision guard's 500-update interval for the usual K
CPU_SAMPLE_MAX_BATCH = 8192   # bounded sample: larger batches are timed on this many trajectories


def host_cpu_info():
    """Model string, sockets, physical cores and logical CPUs of this host (/proc/cpuinfo)."""
    model, phys, cores = None, set(), set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model is None:
                model = v
            elif k == "physical id":
                pid = v
                phys.add(v)
            elif k == "core id":
                cid = v
            elif k == "" and pid is not None and cid is not None:
                cores.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    import platform
    return {"cpu_model": model or platform.processor() or "unknown", "sockets": len(phys) or None,
            "physical_cores": len(cores) or None, "logical_cpus": os.cpu_count()}


def reference_root():
    """The unmodified reference tree, when this box has it (the build container: /root/reference; anywhere else through
    GOPS_REFERENCE_ROOT).  The GPU box of the driver does not: the baseline is then the oracle port (`kind: port`)."""
    for cand in (os.environ.get("GOPS_REFERENCE_ROOT"), "/root/reference"):
        if cand and os.path.isfile(os.path.join(cand, "gops", "algorithm", "fhadp.py")):
            return cand
    return None


def cpu_baseline(cfg, seed, workload, eager_gpu=False):
    """The reference's CPU trainer arithmetic timed on the host cores of THIS box: the unmodified reference classes
    (`gops.algorithm.fhadp.FHADP._compute_gradient`, `gops.algorithm.infadp.INFADP.local_update`, imported behind the gym /
    tensorboard stub of tests/golden/_ref_import.py) when the reference tree is present - `kind: reference`, nothing estimated -,
    and always the oracle port (`oracle/adp_oracle.py`, pinned to the reference's fixtures).  Where the reference is absent the
    stated `value` is the port's rate divided by the reference / port time ratio measured on identical cores in the build
    container (profiles/cpu_port_calibration.json): `kind: port`, `estimated: true`.  Checker only: nothing it computes is used by
    the GPU path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import reference_init_nets
    from oracle import adp_oracle as orc
    B = min(cfg["batch"], CPU_SAMPLE_MAX_BATCH)
    full_batch = CONFIGS[workload]["batch"]
    cfg = dict(cfg, batch=B)
    nets = reference_init_nets(cfg, seed, obs_dim_of(cfg), act_dim_of(cfg))
    env = orc.make_env(cfg["env_id"], pre_horizon=cfg.get("pre_horizon", 10), lq_config=cfg.get("lq_config", "s4a2"))
    data = make_batch(cfg, seed)
    if cfg["alg"] == "FHADP":
        def one(e=env, n=nets, d=data):
            orc.fhadp_gradient(e, n["policy"], d, cfg["horizon"], cfg["gamma"])
        calls_per_unit = 1
    else:   # INFADP: one PEV + one PIM gradient = two bench steps
        def one(e=env, n=nets, d=data):
            orc.infadp_pev_gradient(e, n["policy"], n["v"], n["v_target"], d, cfg["horizon"], cfg["gamma"])
            orc.infadp_pim_gradient(e, n["policy"], n["v_target"], d, cfg["horizon"], cfg["gamma"])
        calls_per_unit = 2
    ref_one, ref_root = None, reference_root()
    if ref_root is not None:   # the unmodified classes
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import _ref_import
        _ref_import.REFERENCE_ROOT = ref_root
        with contextlib.redirect_stdout(sys.stderr):
            import make_golden as mg   # (alg_kwargs / build_alg of the fixture generator: constructs the reference classes)
            ref_alg = mg.build_alg(cfg, seed)
        if cfg["alg"] == "FHADP":
            def ref_one():
                ref_alg._compute_gradient({k: v.clone() for k, v in data.items()})
        else:
            it = [0]

            def ref_one():   # one PEV + one PIM `local_update` (incl. the reference's Adam / Polyak steps: negligible)
                for _ in range(2):
                    ref_alg.local_update({k: v.clone() for k, v in data.items()}, it[0])
                    it[0] += 1
    # The reference pins 4 intra-op threads for serial trainers (gops/utils/init_args.py:31-35); the many tiny ATen ops of
    # this loop scale poorly, so a few counts are tried: the 4-thread figure is reported next to the best one.
    host = host_cpu_info()
    ncpu = host["logical_cpus"] or 4
    steps_per_call = B * cfg["horizon"]

    def sweep(fn):
        res = {}
        for nthreads in sorted({4, min(8, ncpu), min(16, ncpu), min(32, ncpu), min(64, ncpu)}):
            torch.set_num_threads(nthreads)
            times = []
            for i in range(4):
                t0 = time.perf_counter()
                fn()
                times.append((time.perf_counter() - t0) / calls_per_unit)
            res[nthreads] = min(times[1:])
        return res
    port = sweep(one)
    ref = sweep(ref_one) if ref_one is not None else None
    timed = ref if ref is not None else port
    best_threads = min(timed, key=timed.get)
    eager = None
    if eager_gpu and cfg["alg"] == "FHADP":   # the same restatement as plain PyTorch-ROCm eager ops on the GPU ("no-kernel" baseline)
        dev = torch.device("cuda", torch.cuda.current_device())
        src = nets["policy"]
        pol = dict(src, w=[w.detach().to(dev).requires_grad_(True) for w in src["w"]],
                   b=[b.detach().to(dev).requires_grad_(True) for b in src["b"]],
                   act_high=src["act_high"].to(dev), act_low=src["act_low"].to(dev))
        ddata = {k: v.to(dev) for k, v in data.items()}
        def on_dev(x):
            if torch.is_tensor(x):
                return x.to(dev)
            return {k: on_dev(v) for k, v in x.items()} if isinstance(x, dict) else x
        denv = on_dev(env)
        times = []
        for i in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            orc.fhadp_gradient(denv, pol, ddata, cfg["horizon"], cfg["gamma"])
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        eager = steps_per_call / min(times[1:])
    out = {"value": steps_per_call / timed[best_threads], "unit": "env-model steps/s", "cores": best_threads,
           "kind": "reference" if ref is not None else "port", "estimated": False,
           "cpu_model": host["cpu_model"], "sockets": host["sockets"], "physical_cores": host["physical_cores"],
           "logical_cpus": host["logical_cpus"],
           "threads_4": {"value": steps_per_call / timed[4], "threads": 4,
                         "note": "the reference's default intra-op thread count for serial trainers (gops/utils/init_args.py:31-35)"},
           "best": {"value": steps_per_call / timed[best_threads], "threads": best_threads},
           "by_threads": {str(n): steps_per_call / t for n, t in timed.items()},
           "port_by_threads": {str(n): steps_per_call / t for n, t in port.items()},
           "eager_gpu_steps_per_s": eager,
           "sample": f"{B} of the workload's {full_batch} trajectories x H={cfg['horizon']}; per thread count 1 warm-up + 3 "
                     f"timed gradient evaluations (fwd+bwd{', PEV and PIM averaged' if calls_per_unit == 2 else ''}), best; "
                     + ", ".join(f"{n} threads: {t * 1e3:.0f} ms" for n, t in timed.items())
                     + f"; host: {host['cpu_model']}, {host['sockets']} socket(s), {host['physical_cores']} physical cores, "
                       f"{host['logical_cpus']} logical CPUs"}
    if ref is not None:
        out["timed"] = "the UNMODIFIED reference classes from " + ref_root + " (use_gpu=False)"
        out["port_value"] = steps_per_call / port[best_threads]
        out["reference_over_port_time_ratio_here"] = {str(n): ref[n] / port[n] for n in ref}
        return out
    # No reference tree on this box: `value` = the port's rate divided by the reference / port time ratio measured on identical
    # cores in the build container (tools/calibrate_cpu_port.py -> profiles/cpu_port_calibration.json: the port skips the
    # reference's deepcopy(data), per-key info clones and torch.equal host checks); `port_value` keeps what was timed here.
    out["timed"] = "the oracle port (oracle/adp_oracle.py); the reference tree is not on this box"
    cal_path = os.path.join(ROOT, "profiles", "cpu_port_calibration.json")
    if os.path.exists(cal_path):
        cal = json.load(open(cal_path))
        rec = cal.get("workloads", {}).get(workload)
        if rec is not None:
            ths = {int(k): v["ratio_ref_over_port"] for k, v in rec.get("threads", {}).items()}
            ratio = max(ths.values()) if ths else rec["ratio_ref_over_port"]   # the largest measured ratio: the most conservative estimate
            for key in ("value",):
                out["port_value"] = out[key]
                out[key] = out["port_value"] / ratio
            out["threads_4"]["value"] /= ratio
            out["best"]["value"] /= ratio
            out["by_threads"] = {k: v / ratio for k, v in out["by_threads"].items()}
            out["estimated"] = True
            out["calibration"] = {"reference_over_port_time_ratio": ratio,
                                  "ratio_by_threads": {str(k): v for k, v in sorted(ths.items())},
                                  "ratio_spread": [min(ths.values()), max(ths.values())] if ths else None,
                                  "measured_on": f"{cal['cpu']} ({cal['logical_cpus']} logical CPUs: the build container; the largest "
                                                 f"ratio over the measured thread counts is applied)",
                                  "source": "profiles/cpu_port_calibration.json"}
    return out


def pmc_profile(workload, dtype):
    """Per-launch PMC counters of this workload from the newest committed profile
    (profiles/rNN_<workload>[_f16]_pmc_per_launch.json, made by tools/summarize_profile.py from separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes).  None when no profile of THIS workload and
    dtype is committed (another workload's counters are never substituted)."""
    pdir = os.path.join(ROOT, "profiles")
    tag = workload + ("_f16" if dtype == "f16" else "") + "_pmc_per_launch.json"
    files = sorted(f for f in os.listdir(pdir) if f.endswith("_" + tag)) if os.path.isdir(pdir) else []
    if not files:
        return None, None
    return json.load(open(os.path.join(pdir, files[-1]))), "profiles/" + files[-1]


def kernel_bytes(c):
    """HBM bytes of one launch from its counters: FETCH_SIZE (KiB; doubled on gfx950, MI355X_MICROARCH.md) +
    WRITE_SIZE (KiB)."""
    if c is None or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    return 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0


def pmc_traffic(pmc, source, kernel_key):
    """Flat scalars of the dominant kernel from the committed counters: HBM bytes per launch (None without a profile of this
    workload), where they come from, and the share of cycles its matrix pipe was busy."""
    if pmc is None:
        return {"traffic": None, "traffic_bytes": None, "traffic_source": None, "mfma_busy_pct": None}
    # several instantiations can share the name (the value net runs the ENV_NONE one): the rollout's is the largest
    hits = [(kernel_bytes(c), c) for name, c in pmc.items() if kernel_key in name and kernel_bytes(c) is not None]
    if not hits:
        return {"traffic": None, "traffic_bytes": None, "traffic_source": source, "mfma_busy_pct": None}
    nbytes, c = max(hits, key=lambda h: h[0])
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0   # (summed over the 8 XCDs)
    busy = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * gui) if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in c else None
    return {"traffic": nbytes, "traffic_bytes": nbytes, "traffic_source": source, "mfma_busy_pct": busy}


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def relaunch_distributed(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (same contract as the driver's
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`).  On a box with
    fewer than N GPUs the ranks share the devices and exchange gradients over gloo - this exercises the N > 1 code
    path (rendezvous, per-rank shards, collective, deferred 1/N, max-over-ranks timing) and says so in the JSON line."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < args.gpus:
        env.setdefault("GOPS_BENCH_BACKEND", "gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def variant_label(v, dt):
    """gops_rollout_variant bits of the workload's rollouts -> what ran"""
    if v & 1:
        return "register-stationary, plane-split MFMA"
    if v & 4:
        return "streamed, plane-split MFMA"
    if v & 2:
        return "register-stationary, fp32 MFMA"
    if dt != "f32":
        return "streamed (f16 MFMA), 64-trajectory tiles" if v & 8 else "streamed (f16 MFMA), 16-trajectory tiles"
    return "streamed (fp32 MFMA)"


def roofline_of(cfg, dt, kern, pmc, pmc_src, workload, variant=0):
    """Roofline record of the slowest of the three rollout kernels (forward, sweep, weight-gradient group).

    fp32 workloads on the exact-fp32 kernels are priced against the fp32 matrix roof (157.3 TF) with the ALGORITHMIC flops
    2 * MAC.  When the plane-split kernels run (`variant` bits 0 / 2: 3 f16 MFMAs per 32-deep block - two half planes per
    operand - instead of 8 fp32 MFMAs, fp32-class results; the weight-gradient GEMM: 3 bf16 MFMAs) `frac` is the ISSUED fraction -
    3 x the algorithmic flops against the 2.5 PF dense bf16 / f16 roof - and `frac_vs_fp32_roof` keeps the algorithmic figure against 157.3 TF (how far
    the kernel is above / below what an fp32-MFMA implementation could reach; may exceed 1)."""
    B, H = cfg["batch"], cfg["horizon"]
    tail = cfg["alg"] == "INFADP"
    flops = {0: 2.0 * (mac_per_step(cfg) * B * H + (mac_per_step(cfg, "value") * B if tail else 0)),
             2: 2.0 * mac_per_step(cfg) * B * H}
    flops[1] = flops[0]
    flops.update({3: 2.0 * mac_per_step(cfg, "value") * B, 4: 2.0 * mac_per_step(cfg, "value") * B,
                  5: 2.0 * mac_per_step(cfg, "value") * B})
    bps = bytes_per_step(cfg, dt)
    alg_bytes = {0: 0.5 * bps * B * H, 1: 0.5 * bps * B * H, 2: 0.5 * bps * B * H}   # stash written once / read once / read once
    dom = max((0, 1, 2), key=lambda k: kern[k][0])
    dom_ms = kern[dom][0]
    peak_tf = MFMA_PEAK_TFLOPS[dt]
    if dt == "f32":   # exact-fp32 MFMA: arithmetic intensity 115 FLOP/B >> machine balance -> the matrix pipe binds
        achieved = flops[dom] / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        roofline = {"bound": "mfma", "kernel": KERNEL_NAMES[dom], "achieved": achieved, "peak": peak_tf,
                    "unit": "TFLOP/s", "frac": achieved / peak_tf,
                    **pmc_traffic(pmc, pmc_src, KERNEL_NAMES[dom].split("(")[0]),
                    "algorithmic_flops_per_launch": flops[dom], "algorithmic_bytes_per_launch": alg_bytes[dom], "avg_ms": dom_ms}
        if variant & 5:   # plane-split contractions (stationary: bit 0, streamed: bit 2): 3 MFMAs of 16x16x32 per fp32 block, in the H2 weight-gradient GEMM as well
            # The roof is the one of the instructions the kernel ISSUES (dense bf16 / f16 MFMA, 2.5 PF), `achieved` the issued
            # flops = products_per_mac x the algorithmic ones; the comparison with what an exact-fp32 contraction could reach on
            # this chip (157.3 TF fp32 matrix roof, algorithmic flops) is kept beside it as frac_vs_fp32_roof (can exceed 1).
            mult = 3.0   # (rounds 3-4: 4 in the rollout kernels - bf16x3 activations x bf16 + f16 weights)
            issued = {"tflops": mult * achieved, "peak_tflops": MFMA_PEAK_TFLOPS["f16"], "frac": mult * achieved / MFMA_PEAK_TFLOPS["f16"],
                      "products_per_mac": mult}
            roofline.update({"achieved": issued["tflops"], "peak": issued["peak_tflops"], "frac": issued["frac"],
                             "products_per_mac": mult, "algorithmic_tflops": achieved, "frac_vs_fp32_roof": achieved / peak_tf,
                             # the same three views under explicit names, so that no consumer has to guess which one `achieved` is:
                             "mfma_issued": issued,
                             "algorithmic": {"tflops": achieved, "frac_vs_bf16_f16_roof": achieved / MFMA_PEAK_TFLOPS["f16"],
                                             "frac_vs_fp32_matrix_roof": achieved / peak_tf},
                             "arithmetic": "fp32 results from plane-split MFMAs (two half planes per operand: 22-bit weights and activations, "
                                           "ah*wh + (al*wh + ah*wl) / 2^11; weight-gradient GEMM: exact bf16x3): "
                                           "achieved / peak / frac = ISSUED MFMA flops (products_per_mac x the algorithmic 2 MAC) against the dense "
                                           "bf16 / f16 roof (mfma_issued); `algorithmic` holds the 2 MAC flops against both roofs"})
            roofline["alg_hbm_gbs"] = alg_bytes[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    else:             # half-precision MFMA is 16x faster: the stash traffic binds
        achieved = alg_bytes[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        roofline = {"bound": "hbm", "kernel": KERNEL_NAMES[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    **pmc_traffic(pmc, pmc_src, KERNEL_NAMES[dom].split("(")[0]),
                    "algorithmic_bytes_per_launch": alg_bytes[dom], "avg_ms": dom_ms}
    # the one scalar that says whether the kernel re-reads: counter bytes over algorithmic bytes of the same launch
    roofline["traffic_over_algorithmic"] = (roofline["traffic_bytes"] / alg_bytes[dom]) if roofline.get("traffic_bytes") else None
    return roofline, flops


PARITY_TOL = 1e-4   # north_star: results within 1e-4 (relative) of the CPU fp32 reference


def _num(x):
    return x.item() if hasattr(x, "item") else float(x)


def _rel_l2(a, b):
    a, b = torch.as_tensor(a).double().reshape(-1), torch.as_tensor(b).double().reshape(-1)
    den = b.norm().item()
    return (a - b).norm().item() / (den if den > 0 else 1.0)


def parity_stamp(alg, cfg, workload, device, variant_name):
    """Gradient of the kernels about to be timed (same algorithm object, same variant flags, the workload's full batch) against the
    REFERENCE's values for this workload: tests/golden/big_<workload>.npz holds loss, per-parameter gradient norms and 256 sampled
    entries per parameter recorded from the unmodified reference classes on the seed-0 batch and seed-0 random-init weights
    (tests/golden/make_golden.py golden_big).  Runs before the warm-up (the weights are still the seed-0 initialisation)."""
    path = os.path.join(ROOT, "tests", "golden", "big_" + workload + ".npz")
    if not os.path.exists(path):
        return {"checked": False, "reason": "no reference fixture for this workload"}
    g = np.load(path)
    data = make_batch(cfg, 0)
    if abs(data["obs"].double().sum().item() - float(g["chk/obs_sum"])) > 1e-6:
        return {"checked": False, "reason": "seed-0 batch does not reproduce the fixture's checksum"}
    # The fixture's weights: torch.manual_seed(0), then nn.Linear layers in the reference's construction order (fhadp.py:42-44 the
    # policy; infadp.py:41-45 value net first, targets = copies) - the algorithm object here was seeded by its trainer's rule
    # (`set_seed`: +300 for sync trainers), so the seed-0 initialisation is written into its networks (and stays for the timed run).
    torch.manual_seed(0)
    sizes_p, sizes_v = mlp_sizes(cfg), mlp_sizes(cfg, "value")

    def init_into(module, sizes):
        fresh = [torch.nn.Linear(sizes[i], sizes[i + 1]) for i in range(len(sizes) - 1)]
        with torch.no_grad():
            for layer, src in zip(module.linear_layers(), fresh):
                layer.weight.copy_(src.weight)
                layer.bias.copy_(src.bias)
    with torch.no_grad():
        if cfg["alg"] == "FHADP":
            init_into(alg.networks.policy, sizes_p)
        else:
            init_into(alg.networks.v, sizes_v)
            init_into(alg.networks.policy, sizes_p)
            for tgt, src in ((alg.networks.v_target, alg.networks.v), (alg.networks.policy_target, alg.networks.policy)):
                for a, b in zip(tgt.parameters(), src.parameters()):
                    a.copy_(b)
    pol0 = next(alg.networks.policy.parameters())
    if abs(pol0.detach().double().sum().item() - float(g["chk/policy_w0_sum"])) > 1e-6:
        return {"checked": False, "reason": "seed-0 initialisation does not reproduce the fixture's checksum"}
    data = {k: v.to(device) for k, v in data.items()}

    def compare(prefix, params, loss, ref_loss):
        worst, worst_norm = 0.0, 0.0
        for i, prm in enumerate(params):
            gr = prm.grad.detach()
            got = gr.reshape(-1).cpu()[torch.from_numpy(g[f"{prefix}idx{i}"])]
            worst = max(worst, _rel_l2(got, g[f"{prefix}val{i}"]))
            worst_norm = max(worst_norm, abs(gr.double().norm().item() - float(g[prefix + "norms"][i])) / float(g[prefix + "norms"][i]))
        return {"grad_rel_l2": worst, "grad_norm_rel": worst_norm, "loss_rel": abs(loss - ref_loss) / max(1.0, abs(ref_loss))}

    out = {"checked": True, "tolerance": PARITY_TOL, "variant": variant_name, "fixture": "tests/golden/big_" + workload + ".npz",
           "what": "worst per-parameter rel-L2 over 256 sampled gradient entries, worst per-parameter gradient-norm deviation and the "
                   "loss, against the unmodified reference (CPU fp32) on the seed-0 batch / seed-0 weights of this workload"}
    if cfg["alg"] == "FHADP":
        alg._compute_gradient(data)
        loss = _num(alg.tb_info["Loss/Actor loss-RL iter"])
        out.update(compare("grad/", list(alg.networks.policy.parameters()), loss, float(g["loss"])))
    else:
        # the fixture was recorded with the target value net moved away from the online one (make_golden.perturb_targets)
        saved = [p_.detach().clone() for p_ in alg.networks.v_target.parameters()]
        gen = torch.Generator().manual_seed(0 + 1000)
        with torch.no_grad():
            for p_ in alg.networks.v_target.parameters():
                p_.add_((0.05 * (torch.rand(p_.shape, generator=gen) - 0.5)).to(p_.device))
        if abs(next(alg.networks.v_target.parameters()).detach().double().sum().item() - float(g["chk/vt_w0_sum"])) > 1e-5:
            out = {"checked": False, "reason": "perturbed target net does not reproduce the fixture's checksum"}
        else:
            alg._compute_gradient(data, 0)   # PEV
            pev = compare("pev_grad/", list(alg.networks.v.parameters()), _num(alg.tb_info["Loss/Critic loss-RL iter"]), float(g["pev_loss"]))
            alg._compute_gradient(data, 1)   # PIM
            pim = compare("pim_grad/", list(alg.networks.policy.parameters()), _num(alg.tb_info["Loss/Actor loss-RL iter"]), float(g["pim_loss"]))
            out.update({k: max(pev[k], pim[k]) for k in pev})
            out["pev"], out["pim"] = pev, pim
        with torch.no_grad():
            for p_, s0 in zip(alg.networks.v_target.parameters(), saved):
                p_.copy_(s0)
    if out.get("checked"):
        out["ok"] = bool(out["grad_rel_l2"] < PARITY_TOL and out["grad_norm_rel"] < PARITY_TOL and out["loss_rel"] <= PARITY_TOL)
    torch.cuda.synchronize()
    return out



def run_workload(workload, dtype, steps, warmup, profile_steps, ctx, flags=0, check_parity=False, overlap=True, label=None,
                 strict_refpoints=False, dp_path=False, cold_start=False):
    """Times `steps` updates of one workload on this rank's GPU (all ranks call it together) and returns the
    measurements; rank 0 turns them into the record.
    strict_refpoints: the algorithm is built with `strict_reference_points=True` (the appended reference points of every update's
    batch are evaluated on the host with the reference's own torch CPU ops, env/env_ocp/resources/ref_traj_host.py); the loop then
    rotates over four batches and asks for the next batch's points before it queues the current update - what a trainer that
    draws its batches one ahead pays.  dp_path (one rank): the update runs as the data-parallel trainers run it -
    get_remote_update_info (two-phase backward, the collectives' start points) -> average_ -> remote_update - with the
    collectives themselves left out: the fixed cost of that path against the fused local update."""
    rank, world, device, dist = ctx["rank"], ctx["world"], ctx["device"], ctx["dist"]
    cfg = CONFIGS[workload]
    torch.manual_seed(0)   # identical random-init weights on every replica
    # kernel variants travel in the descriptors (GopsRolloutDesc.variant_flags); the algorithm classes build theirs from this default
    saved_flags, hb.DEFAULT_VARIANT_FLAGS = hb.DEFAULT_VARIANT_FLAGS, flags
    with contextlib.redirect_stdout(sys.stderr):   # stdout carries exactly one JSON line
        alg = create_alg(**alg_kwargs(cfg, 0), mlp_dtype=dtype, strict_reference_points=strict_refpoints)
    alg.networks.to(device)
    if cfg["alg"] == "INFADP":   # cfg3 / cfg5: one step = one local_update, PEV and PIM alternate
        alg.gamma, alg.forward_step = cfg["gamma"], cfg["horizon"]
    parity = None
    if check_parity and rank == 0 and dtype == "fp32":
        parity = parity_stamp(alg, cfg, workload, device, label or ("exact fp32 MFMA (GOPS_VF_STREAMED_FP32 | GOPS_VF_DW_F32)" if flags else "default"))
    data = {k: v.to(device) for k, v in make_batch(cfg, 1000 + rank).items()}   # per-rank shard
    rotation = [data] + [{k: v.to(device) for k, v in make_batch(cfg, 2000 + 10 * j + rank).items()} for j in range(3)] if strict_refpoints else None
    reducer = GradAllReducer(overlap=overlap, single_rank_phases=dp_path)   # (overlap=False: one flat all-reduce behind the whole backward)
    if rotation is not None:
        alg.prefetch_reference_points(rotation[0])

    def step(it):
        if rotation is not None:   # a fresh batch per update: its points were asked for one update ago, the next one's are asked for now
            cur = rotation[it % len(rotation)]
            alg.prefetch_reference_points(rotation[(it + 1) % len(rotation)])
            alg.local_update(cur, it)
        elif world == 1 and not dp_path:
            alg.local_update(data, it)
        else:
            if getattr(alg, "supports_overlapped_reduce", False):   # the all-reduce of the early gradients overlaps the rest of the backward
                _, info = alg.get_remote_update_info(data, it, reducer=reducer)
            else:
                _, info = alg.get_remote_update_info(data, it)   # no host sync: the loss stays on the device
            reducer.average_(info, defer_scale=True)         # flat SUM all-reduce (or the wait for the started ones); 1/N applied inside the Adam kernel
            alg.remote_update(info)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Two things stay out of the K timed steps, both measured on the GPU box (tools/gpu/scratch):
    # * the interpreter's cyclic garbage collector (what `timeit` does as well): a generation-2 pass over this process's heap takes
    #   60 - 110 ms on the GPU host (`gc.callbacks`) - ten times a 20-step timed region; it hit one in two short runs and turned a
    #   0.46 ms update into a "4.8 ms" one.  Collected BEFORE any GPU work of this workload, disabled until the timed steps are over;
    # * the GPU's clock ramp: after >= 10 ms of idleness the same 20 updates take 0.51 - 0.52 ms each instead of 0.455 ms (and the
    #   kernels' own HIP-event times move with them), and ~40 updates pass before the steady state - W = 5 warm-up steps are 2.5 ms.
    #   PREHEAT_UPDATES untimed updates run IN FRONT of the W warm-up steps, with no host work between them and the timed region.
    # The timed region still is exactly K complete updates behind W warm-up updates; this only decides at which clocks they run.
    import gc
    gc.collect()
    gc.disable()
    preheat = 0 if (strict_refpoints or os.environ.get("GOPS_BENCH_PREHEAT", "1") == "0") else PREHEAT_UPDATES
    cold_elapsed = None
    if cold_start and preheat:   # the same W + K protocol from a GPU that has just been idle: reported beside `value`, never as `value`
        for it in range(warmup):
            step(it)
        barrier()
        t0 = time.perf_counter()
        for it in range(steps):
            step(warmup + it)
        barrier()
        cold_elapsed = time.perf_counter() - t0
        if world > 1:
            tc = torch.tensor([cold_elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            cold_elapsed = tc.item()
    for it in range(preheat):
        step(it)
    for it in range(warmup):
        step(preheat + it)
    barrier()
    t0 = time.perf_counter()
    for it in range(steps):
        step(preheat + warmup + it)
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    # Per-kernel durations (roofline): HIP events around each launch on the launch stream.  Events
    # cannot be read back from inside a replayed graph, so the same steps are issued once more as
    # plain launches (same kernels, same batch, same stream) right after the timed region.
    graph_mode = os.environ.get("GOPS_HIP_GRAPH")
    os.environ["GOPS_HIP_GRAPH"] = "0"
    hb.profile_reset()
    hb.profile_enable(True)
    for it in range(profile_steps):
        step(preheat + warmup + steps + it)
    barrier()
    hb.profile_enable(False)
    if graph_mode is None:
        del os.environ["GOPS_HIP_GRAPH"]
    else:
        os.environ["GOPS_HIP_GRAPH"] = graph_mode
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    kern = {k: hb.profile_read(k) for k in range(6)}
    hb.profile_reset()
    # the networks after warmup + steps + profile updates on the fixed synthetic batch: still finite?  (outside the timed region; a
    # half-range overflow of the plane-split kernels poisons gradients with NaN on purpose - it must not go unnoticed in a bench run)
    finite = all(bool(torch.isfinite(p_).all()) for p_ in alg.networks.parameters())
    variant = 0
    for ro in list(getattr(alg, "_rollouts", {}).values()) + [o for o in getattr(alg, "_cache", {}).values() if hasattr(o, "desc")]:
        variant |= max(0, hb.lib().gops_rollout_variant(ro.desc))
    payload = None
    if strict_refpoints and alg._reference_pipeline() is not None:
        alg._reference_pipeline().close()
    if world > 1:   # what one update all-reduces: the flat gradient buffer(s) of the network(s) it trains
        payload = {name: sum(p.numel() for p in net.parameters()) * 4 for name, net in alg.networks.net_dict.items()} \
            if hasattr(alg.networks, "net_dict") else {"policy": sum(p.numel() for p in alg.networks.policy.parameters()) * 4}
    evaluated = None
    if strict_refpoints:
        pipe = alg._reference_pipeline()
        evaluated = None if pipe is None else pipe.evaluated
    del alg, data, rotation
    torch.cuda.empty_cache()
    hb.DEFAULT_VARIANT_FLAGS = saved_flags
    return {"elapsed": elapsed, "kern": kern, "variant": variant, "parity": parity, "allreduce_payload_bytes": payload, "finite": finite,
            "refpoint_batches_evaluated": evaluated, "preheat": preheat, "cold_elapsed": cold_elapsed}


def record_of(workload, dtype, steps, warmup, world, m):
    cfg = CONFIGS[workload]
    B, H = cfg["batch"], cfg["horizon"]
    dt = "f16" if dtype == "fp16" else "f32"
    elapsed, kern = m["elapsed"], m["kern"]
    value = world * B * H * steps / elapsed
    pmc, pmc_src = pmc_profile(workload, dt)
    roofline, flops = roofline_of(cfg, dt, kern, pmc, pmc_src, workload, m.get("variant", 0))
    peak_tf = MFMA_PEAK_TFLOPS[dt]
    bps = bytes_per_step(cfg, dt)
    # whole-update fractions of both roofs (SURVEY 8d asks for both next to each other)
    per_step_flops = 6.0 * mac_per_step(cfg)
    hbm_measured, update_bytes = None, None
    if pmc is not None:   # counter bytes of every kernel of one update / the update's time
        tot = [kernel_bytes(c) * c.get("launches_per_update", 1.0) for c in pmc.values() if kernel_bytes(c) is not None]
        if tot:
            update_bytes = sum(tot)
            hbm_measured = update_bytes / (elapsed / steps) / 1e9 / HBM_PEAK_GBS
    split = bool(m.get("variant", 0) & 5)
    arithmetic = ("f16 storage + MFMA, fp32 accumulate/env/results" if dt == "f16" else
                  "2xf16 planes (22-bit operands), fp32 accumulate" if split else "fp32 MFMA (v_mfma_f32_16x16x4_f32)")
    alg_name = cfg["alg"]
    return {
        "metric": f"env-model steps/sec (batch x H), {alg_name} compute_gradient + update",
        "value": value, "unit": "env-model steps/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "preheat_updates": m.get("preheat"),   # untimed, in front of the warm-up steps (clock ramp; GOPS_BENCH_PREHEAT=0: none)
        "cold_start": None if m.get("cold_elapsed") is None else {
            "value": world * B * H * steps / m["cold_elapsed"], "ms_per_step": m["cold_elapsed"] / steps * 1e3,
            "what": "the same W warm-up + K timed steps measured first, on a GPU that had been idle (no pre-heat): the clock ramp of the "
                    "first ~20 ms of load is inside this figure; `value` is the steady state a training run sees"},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dt,
        "arithmetic": arithmetic,
        "data": "synthetic (seeded initial states, random-init networks)",
        "config": {"workload": workload, "env_id": cfg["env_id"], "algorithm": alg_name,
                   "batch_per_gpu": B, "horizon": H, "policy_mlp": mlp_sizes(cfg),
                   "activation": cfg["act"], "parallelism": f"dp{world}"},
        "rollouts_per_sec": world * B * steps / elapsed,
        "roofline": roofline,
        "flops_fraction": value / world * per_step_flops / (peak_tf * 1e12),
        "alg_hbm_fraction": value / world * bps / (HBM_PEAK_GBS * 1e9),
        "hbm_fraction": hbm_measured,
        "hbm_fraction_source": pmc_src if hbm_measured is not None else None,
        "update_traffic_bytes": update_bytes,                        # counters, every kernel of one update
        "update_algorithmic_bytes": 1.5 * bps * B * H,               # stash written once, read by the sweep, read by the weight-gradient GEMMs
        "update_traffic_over_algorithmic": (update_bytes / (1.5 * bps * B * H)) if update_bytes else None,
        "kernels_ms": {KERNEL_NAMES[k]: {"avg_ms": kern[k][0], "launches": kern[k][1],
                                         "tflops": (flops[k] / (kern[k][0] * 1e-3) / 1e12) if kern[k][0] > 0 else 0.0}
                       for k in kern if kern[k][1] > 0},
        "kernel_variant": variant_label(m.get("variant", 0), dt),
        "weights_finite_after_run": m.get("finite"),
        "host_sync_per_step": os.environ.get("GOPS_EAGER_LOG", "0") not in ("", "0"),
        "parity": m.get("parity"),
    }


HEADLINE = "target_veh3dof_fhadp_b4096_h30"
# every BASELINE.json workload, each as (workload, dtype): timed by the default run so that all of them sit under the
# driver's clock (cfg5 in both arithmetics: BASELINE names the fp16 MFMA path for it)
ALL_WORKLOADS = [("cfg1_idp_fhadp_b64_h10", "fp32"), ("cfg2_idp_fhadp_b4096_h30", "fp32"),
                 ("cfg3_veh3dof_infadp_b8192", "fp32"), ("cfg4_veh3dof_fhadp_b4096_h50", "fp32"),
                 ("cfg5_lq_infadp_b65536", "fp32"), ("cfg5_lq_infadp_b65536", "fp16")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None,
                    help="one BASELINE workload (default: the north_star target as the headline, plus a short timing of every "
                         "other BASELINE workload in the `workloads` field)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp16"],
                    help="arithmetic of the MLP contractions: fp32 (exact, parity path) or fp16 (half-precision MFMA, BASELINE cfg5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dp-path", action="store_true",
                    help="one rank: time the data-parallel update path (get_remote_update_info with the two-phase backward -> average_ -> "
                         "remote_update) without collectives instead of the fused local update")
    ap.add_argument("--strict-refpoints", action="store_true",
                    help="build the algorithm with strict_reference_points=True (host-evaluated appended reference points, one batch ahead)")
    ap.add_argument("--no-other-workloads", action="store_true")
    ap.add_argument("--eager-gpu-baseline", action=argparse.BooleanOptionalAction, default=True,
                    help="also time the oracle restatement as PyTorch eager ops on the GPU (FHADP workloads; ~1 s; reported "
                         "inside cpu_baseline as eager_gpu_steps_per_s)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_distributed(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dev_index = local_rank % torch.cuda.device_count()   # (== local_rank on a node with >= N GPUs)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist, backend = None, None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") is the product path; gloo only exists to exercise this script's multi-rank logic on a box
        # with fewer GPUs than ranks (the ranks then share devices)
        backend = os.environ.get("GOPS_BENCH_BACKEND", "nccl" if torch.cuda.device_count() >= world else "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            # gloo's C++ side prints its connection report on stdout, which carries exactly one JSON line: send the
            # process's fd 1 to stderr while the group forms
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(backend)
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
    ctx = {"rank": rank, "world": world, "device": device, "dist": dist}

    workload = args.workload or HEADLINE
    m = run_workload(workload, args.dtype, args.steps, args.warmup, min(args.steps, 100), ctx, check_parity=True,
                     dp_path=args.dp_path and world == 1, strict_refpoints=args.strict_refpoints, cold_start=True)
    out = record_of(workload, args.dtype, args.steps, args.warmup, world, m) if rank == 0 else None
    if world > 1:
        # N > 1: say what carried the gradients, time the collective alone, and time the update loop a second time with the
        # all-reduce NOT overlapped (one flat collective behind the whole backward) - so that a scaling shortfall can be
        # attributed: collective latency (allreduce_us), lost overlap (overlap.on vs .off) or something else
        payload = m["allreduce_payload_bytes"]
        nbytes = max(payload.values())
        buf = torch.zeros(nbytes // 4, dtype=torch.float32, device=device)
        for _ in range(10):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(100):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        t_ar = torch.tensor([(time.perf_counter() - t0) / 100 * 1e6], dtype=torch.float64, device=device)
        dist.all_reduce(t_ar, op=dist.ReduceOp.MAX)
        m_off = run_workload(workload, args.dtype, args.steps, args.warmup, 0, ctx, overlap=False)
        if rank == 0:
            cfg = CONFIGS[workload]
            out["backend"] = "nccl (RCCL)" if backend == "nccl" else f"{backend} - {world} ranks on {torch.cuda.device_count()} GPU(s): multi-rank logic only, not a scaling number"
            out["rccl_ranks"] = world if backend == "nccl" else 0
            out["gpus_visible"] = torch.cuda.device_count()
            out["allreduce_payload_bytes"] = payload
            out["allreduce_us"] = {"value": t_ar.item(), "bytes": nbytes, "calls": 100,
                                   "what": "stand-alone in-place SUM all-reduce of one update's gradient buffer, back to back, max over ranks"}
            out["overlap"] = {"on": {"value": out["value"], "ms_per_step": out["ms_per_step"]},
                              "off": {"value": world * cfg["batch"] * cfg["horizon"] * args.steps / m_off["elapsed"],
                                      "ms_per_step": m_off["elapsed"] / args.steps * 1e3},
                              "what": "the same timed loop with GradAllReducer(overlap=True): FHADP starts the all-reduce of the output / upper "
                                      "layers' gradients behind backward phase A while phase B still runs (headline) - and with overlap=False: "
                                      "one flat all-reduce behind the whole backward"}
    if args.workload is None and world == 1 and not args.no_other_workloads:
        # the other BASELINE workloads, ~1 s each: same timing method, fewer steps
        others = {}

        def brief(r):
            return {"value": r["value"], "ms_per_step": r["ms_per_step"], "steps": r["steps"], "dtype": r["dtype"],
                    "arithmetic": r["arithmetic"],
                    "roofline": {k: r["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_ms", "frac_vs_fp32_roof",
                                                               "products_per_mac", "traffic", "traffic_over_algorithmic", "mfma_busy_pct")
                                 if k in r["roofline"]},
                    "kernel_variant": r["kernel_variant"], "parity": r.get("parity"), "weights_finite_after_run": r.get("weights_finite_after_run"),
                    "kernels_ms": {k: v["avg_ms"] for k, v in r["kernels_ms"].items()}}
        k_steps, k_warm = min(args.steps, 20), min(args.warmup, 5)
        for name, dtype in ALL_WORKLOADS:
            mm = run_workload(name, dtype, k_steps, k_warm, min(k_steps, 10), ctx, check_parity=(dtype == "fp32"))
            others[name + ("_f16" if dtype == "fp16" else "")] = brief(record_of(name, dtype, k_steps, k_warm, world, mm))
        # the headline workload once more on the exact-fp32 kernels (v_mfma_f32_16x16x4_f32 in the rollout kernels and the
        # weight-gradient GEMM): what a strict-fp32 reader takes as the figure, and the parity distance of THAT arithmetic
        mm = run_workload(workload, "fp32", k_steps, k_warm, min(k_steps, 10), ctx, flags=hb.VF_STREAMED_FP32 | hb.VF_DW_F32, check_parity=True)
        others["exact_fp32"] = dict(brief(record_of(workload, "fp32", k_steps, k_warm, world, mm)), workload=workload,
                                    variant_flags="GOPS_VF_STREAMED_FP32 | GOPS_VF_DW_F32")
        # ... and on the register-stationary exact-fp32 kernels (weights resident like the default, every product an fp32 MFMA): the
        # fastest strict-fp32 configuration of this library at this shape
        mm = run_workload(workload, "fp32", k_steps, k_warm, min(k_steps, 10), ctx, flags=hb.VF_NO_STATIONARY_SPLIT | hb.VF_NO_STREAMED_SPLIT_FWD
                          | hb.VF_NO_STREAMED_SPLIT_BWD | hb.VF_DW_F32, check_parity=True, label="exact fp32 MFMA, register-stationary weights")
        others["exact_fp32_stationary"] = dict(brief(record_of(workload, "fp32", k_steps, k_warm, world, mm)), workload=workload,
                                               variant_flags="GOPS_VF_NO_STATIONARY_SPLIT | _NO_STREAMED_SPLIT_FWD | _NO_STREAMED_SPLIT_BWD | GOPS_VF_DW_F32")
        # what the algorithm classes' PrecisionGuard (algorithm/base.py) falls back to when its measured rule trips: exact-fp32 rollout
        # kernels (forward and sweep), the two-half-plane weight-gradient GEMM behind them - and what the guard itself costs while it does not trip
        from gops_amd.algorithm.base import PrecisionGuard
        mm = run_workload(workload, "fp32", k_steps, k_warm, min(k_steps, 10), ctx, flags=PrecisionGuard.exact_rollout_flags(), check_parity=True,
                          label="exact-fp32 rollout kernels (forward and sweep), two-half-plane weight-gradient GEMM")
        exact_ro = brief(record_of(workload, "fp32", k_steps, k_warm, world, mm))
        g = PrecisionGuard()
        others["guard_fallback"] = dict(exact_ro, workload=workload,
                                       variant_flags="GOPS_VF_NO_STATIONARY_SPLIT | GOPS_VF_NO_STREAMED_SPLIT_FWD | GOPS_VF_NO_STREAMED_SPLIT_VALUE")
        out["precision_guard"] = {
            "interval": g.interval, "threshold": g.threshold,
            "checks_in_timed_loop": 0 if g.interval <= 0 else (args.warmup + args.steps) // g.interval - args.warmup // g.interval,
            "amortized_cost_fraction": ((out["ms_per_step"] + exact_ro["ms_per_step"]) / (g.interval * out["ms_per_step"])) if g.interval > 0 else 0.0,
            "what": "every `interval` updates the algorithm classes form the batch's gradient twice more (own kernels, exact-fp32 rollout kernels) and "
                    "compare; the bench loop times compute_gradient + Adam without it - amortized_cost_fraction = (t_default + t_guard_fallback) "
                    "/ (interval * t_default) is what a training run pays on top of `value`"}
        # strict reference points (opt-in parity mode): the appended points of every update's batch from the host's torch
        # CPU ops, a fresh batch per update, the next batch's points evaluated on a side thread while the GPU works on this one -
        # rate and parity stamp of the target and of cfg3 (whose default stamp is the appended-heading effect)
        for name, key in ((workload, "target_strict_refpoints"), ("cfg3_veh3dof_infadp_b8192", "cfg3_strict_refpoints")):
            mm = run_workload(name, "fp32", k_steps, k_warm, 0, ctx, check_parity=True, strict_refpoints=True,
                              label="default kernels, strict_reference_points=True")
            cfgn = CONFIGS[name]
            others[key] = {"workload": name, "value": cfgn["batch"] * cfgn["horizon"] * k_steps / mm["elapsed"],
                           "ms_per_step": mm["elapsed"] / k_steps * 1e3, "steps": k_steps, "parity": mm["parity"],
                           "refpoint_batches_evaluated": mm["refpoint_batches_evaluated"],
                           "what": "strict_reference_points=True: [B, H, 4] appended reference points per update from the host "
                                   "(torch CPU ops = the reference's own values), prefetched one batch ahead; the update is host-bound"}
        # the data-parallel update path on ONE rank, collectives left out: get_remote_update_info (two-phase backward) -> average_ ->
        # remote_update against the fused local update of the headline - the path's fixed cost, so that an N > 1 efficiency is
        # attributable to the collective alone
        mm = run_workload(workload, "fp32", k_steps, k_warm, min(k_steps, 10), ctx, dp_path=True)
        others["dp_path_n1"] = dict(brief(record_of(workload, "fp32", k_steps, k_warm, world, mm)), workload=workload,
                                    what="the N > 1 update path (two-phase backward, average_, remote_update) on one rank without collectives")
        others["dp_path_n1"]["vs_headline"] = others["dp_path_n1"]["value"] / out["value"]
        out["strict_fp32_value"] = others["exact_fp32_stationary"]["value"]   # every product an fp32 MFMA (workloads.exact_fp32_stationary)
        out["workloads"] = others
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(CONFIGS[workload], 0, workload, args.eager_gpu_baseline)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
