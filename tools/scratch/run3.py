import sys, torch, contextlib
sys.path.insert(0, '/root/repo')
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
