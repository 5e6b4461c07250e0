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


CPU_SAMPLE_MAX_BATCH = 8192   # bounded sample: larger batches are timed on this many trajectories


def cpu_baseline(cfg, seed, workload, eager_gpu=False):
    """The oracle (CPU restatement of the reference, pinned to its fixtures) timed on the host
    cores of this box.  Checker only: nothing it computes is used by the GPU path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import reference_init_nets
    from oracle import adp_oracle as orc
    B = min(cfg["batch"], CPU_SAMPLE_MAX_BATCH)
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
    # The reference pins 4 intra-op threads for serial trainers (gops/utils/init_args.py:31-35); the
    # many tiny ATen ops of this loop scale poorly, so try a few counts and report the fastest.
    ncpu = os.cpu_count() or 4
    results = {}
    for nthreads in sorted({4, min(16, ncpu), min(64, ncpu)}):
        torch.set_num_threads(nthreads)
        times = []
        for i in range(3):
            t0 = time.perf_counter()
            one()
            times.append((time.perf_counter() - t0) / calls_per_unit)
        results[nthreads] = min(times[1:])
    best_threads = min(results, key=results.get)
    best = results[best_threads]
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
        eager = B * cfg["horizon"] / min(times[1:])
    out = {"eager_gpu_steps_per_s": eager, "value": B * cfg["horizon"] / best, "unit": "env-model steps/s",
           "cores": best_threads, "kind": "port",
           "sample": f"{B} of the workload's {CONFIGS[workload]['batch']} trajectories x H={cfg['horizon']}; per thread count 1 warm-up + 2 "
                     f"timed gradient evaluations (fwd+bwd{', PEV and PIM averaged' if calls_per_unit == 2 else ''}), best; "
                     + ", ".join(f"{n} threads: {t * 1e3:.0f} ms" for n, t in results.items())
                     + f"; host has {ncpu} logical CPUs"}
    # the port is faster than the reference classes it restates (no deepcopy(data), no per-key info clones, no
    # torch.equal host checks): profiles/cpu_port_calibration.json (tools/calibrate_cpu_port.py, build container)
    # holds the measured ratio on identical cores
    cal_path = os.path.join(ROOT, "profiles", "cpu_port_calibration.json")
    # The stated baseline (`value`) is the port's rate divided by that ratio = an ESTIMATE of the unmodified reference on this
    # host (the reference tree does not exist on the GPU box); `port_value` keeps what was actually timed here.
    if os.path.exists(cal_path):
        cal = json.load(open(cal_path))
        rec = cal.get("workloads", {}).get(workload)
        if rec is not None:
            ths = {int(k): v["ratio_ref_over_port"] for k, v in rec.get("threads", {}).items()}
            near = min(ths, key=lambda t: abs(t - best_threads)) if ths else None
            ratio = ths[near] if near is not None else rec["ratio_ref_over_port"]
            out["port_value"] = out["value"]
            out["value"] = out["port_value"] / ratio
            out["estimated"] = True
            out["calibration"] = {"reference_over_port_time_ratio": ratio,
                                  "ratio_measured_at_threads": near, "reported_threads": best_threads,
                                  "all_ratios": {str(k): v for k, v in sorted(ths.items())},
                                  "measured_on": f"{cal['cpu']} ({cal['logical_cpus']} logical CPUs: the build container - the ratio at the "
                                                 f"reported thread count cannot be measured there, the nearest measured count is used)",
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
    if pmc is None:
        return None
    # several instantiations can share the name (the value net runs the ENV_NONE one): the rollout's is the largest
    hits = [kernel_bytes(c) for name, c in pmc.items() if kernel_key in name and kernel_bytes(c) is not None]
    return {"bytes": max(hits), "source": source} if hits else None


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
    2 * MAC.  When the plane-split kernels run (`variant` bits 0 / 2: 3 bf16 + 1 f16 MFMA per 32-deep block instead of 8 fp32
    MFMAs, fp32-class results) `frac` is the ISSUED fraction - 4 x the algorithmic flops (3 x in the weight-gradient GEMM)
    against the 2.5 PF dense bf16 / f16 roof - and `frac_vs_fp32_roof` keeps the algorithmic figure against 157.3 TF (how far
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
                    "traffic": pmc_traffic(pmc, pmc_src, KERNEL_NAMES[dom].split("(")[0]),
                    "algorithmic_flops_per_launch": flops[dom], "avg_ms": dom_ms}
        if variant & 5:   # plane-split contractions (stationary: bit 0, streamed: bit 2): 4 MFMAs of 16x16x32 per fp32 block; the H2 weight-gradient GEMM: 3
            # The roof is the one of the instructions the kernel ISSUES (dense bf16 / f16 MFMA, 2.5 PF), `achieved` the issued
            # flops = products_per_mac x the algorithmic ones; the comparison with what an exact-fp32 contraction could reach on
            # this chip (157.3 TF fp32 matrix roof, algorithmic flops) is kept beside it as frac_vs_fp32_roof (can exceed 1).
            mult = 3.0 if dom == 2 else 4.0
            roofline.update({"achieved": mult * achieved, "peak": MFMA_PEAK_TFLOPS["f16"], "frac": mult * achieved / MFMA_PEAK_TFLOPS["f16"],
                             "products_per_mac": mult, "algorithmic_tflops": achieved, "frac_vs_fp32_roof": achieved / peak_tf,
                             "arithmetic": "fp32 results from bf16 / f16 plane-split MFMAs (>= 19-bit weights, exact bf16x3 activations): "
                                           "achieved / peak = issued MFMA flops against the dense bf16 / f16 roof"})
            roofline["alg_hbm_gbs"] = alg_bytes[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    else:             # half-precision MFMA is 16x faster: the stash traffic binds
        achieved = alg_bytes[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        roofline = {"bound": "hbm", "kernel": KERNEL_NAMES[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": pmc_traffic(pmc, pmc_src, KERNEL_NAMES[dom].split("(")[0]),
                    "algorithmic_bytes_per_launch": alg_bytes[dom], "avg_ms": dom_ms}
    return roofline, flops


def run_workload(workload, dtype, steps, warmup, profile_steps, ctx):
    """Times `steps` updates of one workload on this rank's GPU (all ranks call it together) and returns the
    measurements; rank 0 turns them into the record."""
    rank, world, device, dist = ctx["rank"], ctx["world"], ctx["device"], ctx["dist"]
    cfg = CONFIGS[workload]
    torch.manual_seed(0)   # identical random-init weights on every replica
    with contextlib.redirect_stdout(sys.stderr):   # stdout carries exactly one JSON line
        alg = create_alg(**alg_kwargs(cfg, 0), mlp_dtype=dtype)
    alg.networks.to(device)
    if cfg["alg"] == "INFADP":   # cfg3 / cfg5: one step = one local_update, PEV and PIM alternate
        alg.gamma, alg.forward_step = cfg["gamma"], cfg["horizon"]
    data = {k: v.to(device) for k, v in make_batch(cfg, 1000 + rank).items()}   # per-rank shard
    reducer = GradAllReducer(overlap=os.environ.get("GOPS_BENCH_OVERLAP", "1") != "0")   # (A/B knob: 0 = one flat all-reduce behind the whole backward)

    def step(it):
        if world == 1:
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

    for it in range(warmup):
        step(it)
    barrier()
    t0 = time.perf_counter()
    for it in range(steps):
        step(warmup + it)
    barrier()
    elapsed = time.perf_counter() - t0
    # Per-kernel durations (roofline): HIP events around each launch on the launch stream.  Events
    # cannot be read back from inside a replayed graph, so the same steps are issued once more as
    # plain launches (same kernels, same batch, same stream) right after the timed region.
    graph_mode = os.environ.get("GOPS_HIP_GRAPH")
    os.environ["GOPS_HIP_GRAPH"] = "0"
    hb.profile_reset()
    hb.profile_enable(True)
    for it in range(profile_steps):
        step(warmup + steps + it)
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
    variant = 0
    for ro in list(getattr(alg, "_rollouts", {}).values()) + [o for o in getattr(alg, "_cache", {}).values() if hasattr(o, "desc")]:
        variant |= max(0, hb.lib().gops_rollout_variant(ro.desc))
    del alg, data
    torch.cuda.empty_cache()
    return {"elapsed": elapsed, "kern": kern, "variant": variant}


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
    hbm_measured = None
    if pmc is not None:   # counter bytes of every kernel of one update / the update's time
        tot = [kernel_bytes(c) * c.get("launches_per_update", 1.0) for c in pmc.values() if kernel_bytes(c) is not None]
        if tot:
            hbm_measured = sum(tot) / (elapsed / steps) / 1e9 / HBM_PEAK_GBS
    alg_name = cfg["alg"]
    return {
        "metric": f"env-model steps/sec (batch x H), {alg_name} compute_gradient + update",
        "value": value, "unit": "env-model steps/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dt,
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
        "kernels_ms": {KERNEL_NAMES[k]: {"avg_ms": kern[k][0], "launches": kern[k][1],
                                         "tflops": (flops[k] / (kern[k][0] * 1e-3) / 1e12) if kern[k][0] > 0 else 0.0}
                       for k in kern if kern[k][1] > 0},
        "kernel_variant": variant_label(m.get("variant", 0), dt),
        "host_sync_per_step": os.environ.get("GOPS_EAGER_LOG", "0") not in ("", "0"),
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
    m = run_workload(workload, args.dtype, args.steps, args.warmup, min(args.steps, 100), ctx)
    out = record_of(workload, args.dtype, args.steps, args.warmup, world, m) if rank == 0 else None
    if world > 1 and rank == 0:
        out["backend"] = "nccl (RCCL)" if backend == "nccl" else f"{backend} - {world} ranks on {torch.cuda.device_count()} GPU(s): multi-rank logic only, not a scaling number"
    if args.workload is None and world == 1 and not args.no_other_workloads:
        # the other BASELINE workloads, ~1 s each: same timing method, fewer steps
        others = {}
        for name, dtype in ALL_WORKLOADS:
            k_steps, k_warm = min(args.steps, 20), min(args.warmup, 5)
            mm = run_workload(name, dtype, k_steps, k_warm, min(k_steps, 10), ctx)
            r = record_of(name, dtype, k_steps, k_warm, world, mm)
            others[name + ("_f16" if dtype == "fp16" else "")] = {
                "value": r["value"], "ms_per_step": r["ms_per_step"], "steps": k_steps, "dtype": r["dtype"],
                "roofline": {k: r["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_ms", "frac_vs_fp32_roof",
                                                           "products_per_mac") if k in r["roofline"]},
                "kernel_variant": r["kernel_variant"],
                "kernels_ms": {k: v["avg_ms"] for k, v in r["kernels_ms"].items()}}
        out["workloads"] = others
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(CONFIGS[workload], 0, workload, args.eager_gpu_baseline)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
