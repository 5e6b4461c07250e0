"""Time one MPG update (example configuration: cartpoleconti, B = 256, H = 10, 64-64 networks) with and without
HIP-graph replay:   python tools/time_mpg.py"""
import os
import sys
import time

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
from test_alg_gpu import _kwargs  # noqa: E402
from gops_amd.create_pkg.create_alg import create_alg  # noqa: E402
from gops_amd.utils.synthetic import make_batch  # noqa: E402

cfg = dict(alg="MPG", env_id="gym_cartpoleconti", batch=256, horizon=10, hidden=(64, 64), act="relu", gamma=0.99)
extra = dict(pge_method="mixed_weight", eta=0.3, terminal_iter=1e8, forward_step=10, tau=0.1)
for mode in ("0", "1"):
    os.environ["GOPS_HIP_GRAPH"] = mode
    torch.manual_seed(0)
    alg = create_alg(**_kwargs(cfg, extra, 0))
    alg.networks.to("cuda")
    g = torch.Generator().manual_seed(1)
    obs = make_batch(cfg, 3)["obs"].cuda()
    data = dict(obs=obs, act=(torch.rand(256, 1, generator=g) * 2 - 1).cuda(), rew=torch.randn(256, generator=g).cuda(),
                obs2=obs.clone(), done=torch.zeros(256).cuda())
    for it in range(20):
        alg.local_update(data, it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(200):
        alg.local_update(data, 20 + it)
    torch.cuda.synchronize()
    print(f"GOPS_HIP_GRAPH={mode}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per MPG update (incl. the log's host sync)")
