"""Three updates of the target workload (veh3dof FHADP, B=4096, H=30) - the driver for the in-kernel phase
counters:  make -C gops_amd/csrc -B DBG=1 && GOPS_DBG_TIMING=1 python tools/dbg_run.py   (stderr shows cycles / step)."""
import sys, torch, contextlib
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from bench import alg_kwargs
from gops_amd.create_pkg.create_alg import create_alg
from gops_amd.utils.synthetic import CONFIGS, make_batch
cfg = CONFIGS["target_veh3dof_fhadp_b4096_h30"]
with contextlib.redirect_stdout(sys.stderr):
    alg = create_alg(**alg_kwargs(cfg, 0))
alg.networks.to("cuda")
data = {k: v.cuda() for k, v in make_batch(cfg, 1000).items()}
for it in range(3):
    alg.local_update(data, it)
torch.cuda.synchronize()
