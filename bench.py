#!/usr/bin/env python
"""Benchmark of the GOPS ADP hot path on MI355X.

A "step" is one trainer iteration of the hot path on one batch: FHADP `compute_gradient` (fused
forward rollout + backward sweep + weight-gradient GEMMs through libgops_hip.so), the gradient
all-reduce when N > 1, and the Adam update.  Inputs are synthetic and already resident in HBM.
Metric (BASELINE.json): env-model steps/s = N * B * H * K / wall time, weak scaling (per-GPU batch
fixed).  Launch for N > 1:  python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...
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

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
KERNEL_NAMES = {0: "rollout_fwd_kernel", 1: "rollout_bwd_kernel", 2: "dw_gemm_kernel(+reduce)"}


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


def mac_per_step(cfg):
    sizes = [obs_dim_of(cfg) + (1 if cfg["alg"] == "FHADP" else 0)] + list(cfg["hidden"]) + [act_dim_of(cfg)]
    return sum(a * b for a, b in zip(sizes[:-1], sizes[1:]))


def cpu_baseline(cfg, seed, eager_gpu=False):
    """The oracle (CPU restatement of the reference, pinned to its fixtures) timed on the host
    cores of this box.  Checker only: nothing it computes is used by the GPU path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import reference_init_nets
    from oracle import adp_oracle as orc
    nets = reference_init_nets(cfg, seed, obs_dim_of(cfg), act_dim_of(cfg))
    env = orc.make_env(cfg["env_id"], pre_horizon=cfg.get("pre_horizon", 10), lq_config=cfg.get("lq_config", "s4a2"))
    data = make_batch(cfg, seed)
    # The reference pins 4 intra-op threads for serial trainers (gops/utils/init_args.py:31-35); the
    # many tiny ATen ops of this loop scale poorly, so try a few counts and report the fastest.
    ncpu = os.cpu_count() or 4
    results = {}
    for nthreads in sorted({4, min(16, ncpu), min(64, ncpu)}):
        torch.set_num_threads(nthreads)
        times = []
        for i in range(3):
            t0 = time.perf_counter()
            orc.fhadp_gradient(env, nets["policy"], data, cfg["horizon"], cfg["gamma"])
            times.append(time.perf_counter() - t0)
        results[nthreads] = min(times[1:])
    best_threads = min(results, key=results.get)
    best = results[best_threads]
    eager = None
    if eager_gpu:   # the same restatement as plain PyTorch-ROCm eager ops on the GPU ("no-kernel" baseline)
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
        eager = cfg["batch"] * cfg["horizon"] / min(times[1:])
    return {"eager_gpu_steps_per_s": eager, "value": cfg["batch"] * cfg["horizon"] / best, "unit": "env-model steps/s",
            "cores": best_threads, "kind": "port",
            "sample": f"full workload batch (B={cfg['batch']}, H={cfg['horizon']}); per thread count 1 warm-up + 2 "
                      f"timed compute_gradient calls (fwd+bwd), best call; "
                      + ", ".join(f"{n} threads: {t * 1e3:.0f} ms" for n, t in results.items())
                      + f"; host has {ncpu} logical CPUs"}


def pmc_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC pass
    (profiles/rNN_pmc_per_launch.json, made by tools/summarize_profile.py from separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs; FETCH_SIZE doubled per
    MI355X_MICROARCH.md, both counters are KiB).  None when no profile is committed."""
    pdir = os.path.join(ROOT, "profiles")
    files = sorted(f for f in os.listdir(pdir) if f.endswith("_pmc_per_launch.json")) if os.path.isdir(pdir) else []
    if not files:
        return None
    pmc = json.load(open(os.path.join(pdir, files[-1])))
    for name, c in pmc.items():
        if kernel_key in name and "WRITE_SIZE" in c and ("FETCH_SIZE" in c or "TCC_EA0_RDREQ_sum" in c):
            # reads: FETCH_SIZE (KiB, x2 on gfx950) or, where that pass was unavailable, the L2's HBM-side read
            # requests x 128 B (the two agree to 0.1 % where both were collected)
            rd = 2.0 * c["FETCH_SIZE"] * 1024.0 if "FETCH_SIZE" in c else c["TCC_EA0_RDREQ_sum"] * 128.0
            return {"bytes": rd + c["WRITE_SIZE"] * 1024.0, "source": "profiles/" + files[-1]}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="target_veh3dof_fhadp_b4096_h30")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp16"],
                    help="arithmetic of the MLP contractions: fp32 (exact, parity path) or fp16 (half-precision MFMA, BASELINE cfg5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager-gpu-baseline", action="store_true",
                    help="also time the oracle restatement as PyTorch eager ops on the GPU (reported inside cpu_baseline)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev_index = local_rank % torch.cuda.device_count()   # (== local_rank on a node with >= N GPUs)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") is the product path; GOPS_BENCH_BACKEND=gloo only exists to exercise this
        # script's multi-rank logic on a single-GPU box (ranks then share the device)
        backend = os.environ.get("GOPS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    cfg = CONFIGS[args.workload]
    torch.manual_seed(0)   # identical random-init weights on every replica
    with contextlib.redirect_stdout(sys.stderr):   # stdout carries exactly one JSON line
        alg = create_alg(**alg_kwargs(cfg, 0), mlp_dtype=args.dtype)
    alg.networks.to(device)
    if cfg["alg"] == "INFADP":   # cfg3 / cfg5: one step = one local_update, PEV and PIM alternate
        alg.gamma, alg.forward_step = cfg["gamma"], cfg["horizon"]
    data = {k: v.to(device) for k, v in make_batch(cfg, 1000 + rank).items()}   # per-rank shard
    reducer = GradAllReducer()

    def step(it):
        if world == 1:
            alg.local_update(data, it)
        else:
            _, info = alg.get_remote_update_info(data, it)
            reducer.average_(info)
            alg.remote_update(info)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for it in range(args.warmup):
        step(it)
    barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        step(args.warmup + it)
    barrier()
    elapsed = time.perf_counter() - t0
    # Per-kernel durations (roofline): HIP events around each launch on the launch stream.  Events
    # cannot be read back from inside a replayed graph, so the same steps are issued once more as
    # plain launches (same kernels, same batch, same stream) right after the timed region.
    os.environ["GOPS_HIP_GRAPH"] = "0"
    hb.profile_reset()
    hb.profile_enable(True)
    for it in range(min(args.steps, 100)):
        step(args.warmup + args.steps + it)
    barrier()
    hb.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    if rank == 0:
        B, H = cfg["batch"], cfg["horizon"]
        steps_total = world * B * H * args.steps
        kern = {k: hb.profile_read(k) for k in (0, 1, 2)}
        flops_per_launch = 2.0 * mac_per_step(cfg) * B * H      # each of fwd / dX sweep / dW
        dom = max(kern, key=lambda k: kern[k][0])
        dom_ms = kern[dom][0]
        achieved = flops_per_launch / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        out = {
            "metric": "env-model steps/sec (batch x H), FHADP compute_gradient + update",
            "value": steps_total / elapsed, "unit": "env-model steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (seeded initial states, random-init networks)",
            "config": {"workload": args.workload, "env_id": cfg["env_id"], "algorithm": cfg["alg"],
                       "batch_per_gpu": B, "horizon": H, "policy_mlp": [obs_dim_of(cfg) + (1 if cfg["alg"] == "FHADP" else 0)] + list(cfg["hidden"]) + [act_dim_of(cfg)],
                       "activation": cfg["act"], "parallelism": f"dp{world}"},
            "rollouts_per_sec": world * B * args.steps / elapsed,
            "roofline": {"bound": "mfma", "kernel": KERNEL_NAMES[dom], "achieved": achieved,
                         "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                         "traffic": pmc_traffic(KERNEL_NAMES[dom].split("(")[0]), "algorithmic_flops_per_launch": flops_per_launch,
                         "avg_ms": dom_ms},
            "kernels_ms": {KERNEL_NAMES[k]: {"avg_ms": kern[k][0], "launches": kern[k][1],
                                             "tflops": (flops_per_launch / (kern[k][0] * 1e-3) / 1e12) if kern[k][0] > 0 else 0.0}
                           for k in kern},
        }
        if world == 1 and not args.no_cpu_baseline and cfg["alg"] == "FHADP":
            out["cpu_baseline"] = cpu_baseline(cfg, 0, args.eager_gpu_baseline)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
