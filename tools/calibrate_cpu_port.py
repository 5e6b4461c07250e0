#!/usr/bin/env python
"""Calibration of bench.py's `cpu_baseline` (kind "port"): times the UNMODIFIED reference classes
(`gops.algorithm.fhadp.FHADP._compute_gradient`, `gops.algorithm.infadp.INFADP.local_update`, imported from
/root/reference behind the gym / tensorboard stub of tests/golden/_ref_import.py) and the oracle port
(`oracle/adp_oracle.py`, what bench.py times on the GPU box, where /root/reference does not exist) on the
SAME host cores, same batch, same thread counts, and writes profiles/cpu_port_calibration.json:
ratio = reference time / port time per workload.  bench.py quotes the ratio next to its port number.

Runs in the BUILD container only (needs /root/reference); test / measurement infrastructure, not product.
    python tools/calibrate_cpu_port.py [--threads 4 8] [--workloads target_veh3dof_fhadp_b4096_h30 ...]
"""
import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]

import _ref_import  # noqa: E402,F401  (stubs gym/tensorboard, puts /root/reference on sys.path)
import torch  # noqa: E402

import make_golden as mg  # noqa: E402  (alg_kwargs / build_alg of the fixture generator: reference classes)
from helpers import reference_init_nets  # noqa: E402
from oracle import adp_oracle as orc  # noqa: E402

from gops_amd.utils.synthetic import CONFIGS, act_dim_of, make_batch, obs_dim_of  # noqa: E402


def best_of(fn, n=3):
    fn()   # warm-up
    times = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return min(times)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, nargs="*", default=[4, os.cpu_count() or 4])
    ap.add_argument("--workloads", nargs="*", default=["target_veh3dof_fhadp_b4096_h30", "cfg2_idp_fhadp_b4096_h30",
                                                        "cfg3_veh3dof_infadp_b8192", "cfg5_lq_infadp_b65536"])
    ap.add_argument("--max-batch", type=int, default=8192, help="bounded sample: larger workloads are timed on this many trajectories")
    args = ap.parse_args()
    out = {"cpu": cpu_model(), "logical_cpus": os.cpu_count(), "torch": torch.__version__, "workloads": {}}
    for name in args.workloads:
        cfg = dict(CONFIGS[name])
        B = min(cfg["batch"], args.max_batch)
        cfg["batch"] = B
        data = make_batch(cfg, 0)
        ref_alg = mg.build_alg(cfg, 0)
        nets = reference_init_nets(cfg, 0, obs_dim_of(cfg), act_dim_of(cfg))
        env = orc.make_env(cfg["env_id"], pre_horizon=cfg.get("pre_horizon", 10), lq_config=cfg.get("lq_config", "s4a2"))
        rec = {"batch": B, "horizon": cfg["horizon"], "threads": {}}
        for nt in sorted(set(args.threads)):
            torch.set_num_threads(nt)
            if cfg["alg"] == "FHADP":
                t_ref = best_of(lambda: ref_alg._compute_gradient({k: v.clone() for k, v in data.items()}))
                t_port = best_of(lambda: orc.fhadp_gradient(env, nets["policy"], data, cfg["horizon"], cfg["gamma"]))
            else:   # one PEV + one PIM gradient; the reference's local_update also applies Adam / Polyak (negligible)
                it = [0]

                def ref_step():
                    for _ in range(2):
                        ref_alg.local_update({k: v.clone() for k, v in data.items()}, it[0])
                        it[0] += 1

                def port_step():
                    orc.infadp_pev_gradient(env, nets["policy"], nets["v"], nets["v_target"], data, cfg["horizon"], cfg["gamma"])
                    orc.infadp_pim_gradient(env, nets["policy"], nets["v_target"], data, cfg["horizon"], cfg["gamma"])
                t_ref, t_port = best_of(ref_step) / 2, best_of(port_step) / 2
            steps = B * cfg["horizon"]
            rec["threads"][str(nt)] = {"reference_s": t_ref, "port_s": t_port, "ratio_ref_over_port": t_ref / t_port,
                                       "reference_steps_per_s": steps / t_ref, "port_steps_per_s": steps / t_port}
            print(name, f"threads={nt}: reference {t_ref * 1e3:.0f} ms, port {t_port * 1e3:.0f} ms, ratio {t_ref / t_port:.2f}", flush=True)
        rec["ratio_ref_over_port"] = min(v["ratio_ref_over_port"] for v in rec["threads"].values())
        out["workloads"][name] = rec
    path = os.path.join(ROOT, "profiles", "cpu_port_calibration.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
