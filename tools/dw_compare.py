"""Weight gradients of one workload from the default weight-gradient GEMM against the exact three-plane bf16 GEMM
(GOPS_DW_EXACT=1, read at the library's first use: run this script once per setting and diff the saved tensors):
    python tools/dw_compare.py save /tmp/a.pt ; GOPS_DW_EXACT=1 python tools/dw_compare.py save /tmp/b.pt ; python tools/dw_compare.py diff /tmp/a.pt /tmp/b.pt"""
import contextlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "diff":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for i, (x, y) in enumerate(zip(a, b)):
        print(f"tensor {i} {tuple(x.shape)}: rel-L2 {((x - y).double().norm() / y.double().norm()).item():.3e}  max|diff|/max|ref| {((x - y).abs().max() / y.abs().max()).item():.3e}")
    sys.exit(0)
from bench import alg_kwargs  # noqa: E402
from gops_amd.create_pkg.create_alg import create_alg  # noqa: E402
from gops_amd.utils.synthetic import CONFIGS, make_batch  # noqa: E402

workload = sys.argv[3] if len(sys.argv) > 3 else "target_veh3dof_fhadp_b4096_h30"
cfg = CONFIGS[workload]
torch.manual_seed(0)
with contextlib.redirect_stdout(sys.stderr):
    alg = create_alg(**alg_kwargs(cfg, 0))
alg.networks.to("cuda")
data = {k: v.cuda() for k, v in make_batch(cfg, 1000).items()}
os.environ["GOPS_HIP_GRAPH"] = "0"
_, info = alg.get_remote_update_info(data, 0)
torch.cuda.synchronize()
torch.save([g.detach().cpu().clone() for g in info["grad"]], sys.argv[2])
