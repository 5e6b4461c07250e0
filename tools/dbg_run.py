"""A few updates of one workload - the driver for the in-kernel phase counters and the rocprofv3 --pmc passes:
    make -C gops_amd/csrc dbg && GOPS_HIP_LIB=gops_amd/libgops_hip_dbg.so GOPS_DBG_TIMING=1 \
        python tools/dbg_run.py [workload] [fp32|fp16] [updates]
(stderr shows cycles / step of block 0; default: the target workload veh3dof FHADP B=4096 H=30, fp32, 4 updates)."""
import contextlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import alg_kwargs  # noqa: E402
from gops_amd.create_pkg.create_alg import create_alg  # noqa: E402
from gops_amd.utils.synthetic import CONFIGS, make_batch  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "target_veh3dof_fhadp_b4096_h30"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
updates = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg = CONFIGS[workload]
with contextlib.redirect_stdout(sys.stderr):
    alg = create_alg(**alg_kwargs(cfg, 0), mlp_dtype=dtype)
alg.networks.to("cuda")
if cfg["alg"] == "INFADP":
    alg.gamma, alg.forward_step = cfg["gamma"], cfg["horizon"]
data = {k: v.cuda() for k, v in make_batch(cfg, 1000).items()}
for it in range(updates):
    alg.local_update(data, it)
torch.cuda.synchronize()
