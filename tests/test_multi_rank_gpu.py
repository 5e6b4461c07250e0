"""GPU: the data-parallel update path with the REAL algorithm classes, two ranks sharing the one GPU of the test box.

`FHADP / INFADP.get_remote_update_info -> GradAllReducer.average_(defer_scale=True) -> remote_update` is what
`on_sync_trainer` / `off_sync_trainer` / `bench.py --gpus N` run per update (reference semantics:
gops/trainer/on_sync_trainer.py:84-105,189-194 - the N samplers' batches concatenated into one update;
off_sync_trainer.py:183-208 - the mean of N replica gradients).  The collective runs over gloo on CUDA tensors here
(one GPU, two processes); on an 8-GPU node the same code runs over RCCL.  Checked: after several updates every rank
holds the weights a single process reaches with `local_update` on the concatenated 2B batch, and the captured HIP
graph of the gradient kernels keeps replaying between the collectives.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
ROOT, ALG = sys.argv[1], sys.argv[2]
OVERLAP = len(sys.argv) > 4 and sys.argv[4] == "overlap"   # FHADP: the all-reduce of the early gradients overlaps the rest of the backward
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ["GOPS_HIP_GRAPH"] = "1"          # capture even these small steps: the collective sits BETWEEN two graphs
from gops_amd.create_pkg.create_alg import create_alg
from gops_amd.trainer.grad_sync import GradAllReducer, broadcast_parameters
from gops_amd.utils.synthetic import make_batch
from test_alg_gpu import _kwargs

BACKEND = sys.argv[5] if len(sys.argv) > 5 else "gloo"   # "nccl" (= RCCL): one GPU per rank; "gloo": the ranks share GPU 0
local = int(os.environ.get("LOCAL_RANK", 0)) if BACKEND == "nccl" else 0
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group(BACKEND, **({"device_id": dev} if BACKEND == "nccl" else {}))
r, n = dist.get_rank(), dist.get_world_size()
B, ITERS = 96, 7
if ALG == "FHADP":
    cfg = dict(alg="FHADP", env_id="pyth_idpendulum", batch=n * B, horizon=8, hidden=(64, 64), act="gelu", gamma=0.99)
else:
    cfg = dict(alg="INFADP", env_id="pyth_lq", lq_config="s4a2", batch=n * B, horizon=6, hidden=(64, 64), act="elu", gamma=0.99)

def make():
    torch.manual_seed(11)
    alg = create_alg(**_kwargs(dict(cfg, batch=B), {}, 11))
    if ALG == "INFADP":
        alg.forward_step = cfg["horizon"]
    alg.gamma = cfg["gamma"]
    alg.networks.to(dev)
    return alg

alg = make()
broadcast_parameters(alg.networks, src=0)
reducer = GradAllReducer()
full = [{k: v.to(dev) for k, v in make_batch(cfg, 500 + it).items()} for it in range(ITERS)]

def run(alg, overlap):
    for it in range(ITERS):
        shard = {k: v[r * B:(r + 1) * B].contiguous() for k, v in full[it].items()}
        if overlap:   # what the trainers / bench.py do for algorithms with supports_overlapped_reduce
            _, info = alg.get_remote_update_info(shard, it, reducer=reducer)
            assert info.get("_pending") is True and len(reducer._works) == 2
        else:
            _, info = alg.get_remote_update_info(shard, it)
        reducer.average_(info, defer_scale=True)
        assert info["_grad_scale"] == 1.0 / n and not reducer._works
        alg.remote_update(info)
    torch.cuda.synchronize()

run(alg, OVERLAP)
if ALG == "FHADP":
    assert alg._grad_graph.graph is not None, "the gradient kernels were not replayed as a HIP graph"
if OVERLAP:
    # the overlapped path (backward in two halves, two collectives started by the algorithm) against the serial one (one
    # backward, one flat all-reduce): bit-identical weights after the same updates
    assert alg._grad_graph_b.graph is not None
    ser = make()
    broadcast_parameters(ser.networks, src=0)
    run(ser, False)
    for (name, a), b in zip(alg.networks.named_parameters(), ser.networks.parameters()):
        assert torch.equal(a, b), name

# single process, concatenated batch (what the reference's on_sync_trainer feeds its one learner)
os.environ["GOPS_HIP_GRAPH"] = "0"
ref = make()
for it in range(ITERS):
    ref.local_update(full[it], it)
torch.cuda.synchronize()
worst = 0.0
for (name, a), b in zip(alg.networks.named_parameters(), ref.networks.parameters()):
    d = (a - b).abs().max().item() / max(1.0, b.abs().max().item())
    worst = max(worst, d)
    assert d <= 1e-6, (name, d)
moved = max((a - b).abs().max().item() for a, b in zip(make().networks.parameters(), ref.networks.parameters()))
assert moved > 1e-3, moved                       # the updates did move the weights
# every rank ends on the same weights
flat = torch.cat([p.detach().reshape(-1) for p in alg.networks.parameters()])
both = [torch.zeros_like(flat) for _ in range(n)]
dist.all_gather(both, flat)
assert all(torch.equal(both[0], t) for t in both)
# a local_update afterwards steps with plain gradients again (the 1/N of the remote path does not linger)
for opt in alg.networks.optimizer_dict.values():
    assert opt.grad_scale == 1.0
print(f"rank {r} ({BACKEND}, cuda:{local}): {ALG} data-parallel == single-process on the 2B batch, worst rel diff {worst:.2e}", flush=True)
dist.barrier()
dist.destroy_process_group()
open(os.path.join(sys.argv[3], f"ok_{r}"), "w").write("ok")
"""


def _run_two_ranks(tmp_path, alg, port, mode, backend):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, alg, str(tmp_path), mode,
                          backend], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert all((tmp_path / f"ok_{k}").exists() for k in range(2))
    return out.stdout


@pytest.mark.parametrize("alg,port,mode", [("FHADP", 29711, "serial"), ("INFADP", 29712, "serial"), ("FHADP", 29713, "overlap")])
def test_two_ranks_equal_single_process_on_concatenated_batch(tmp_path, alg, port, mode):
    """mode "overlap" (VERDICT r3 #8): the all-reduce of the output / upper hidden layers' gradients is started behind the first
    half of the backward and overlaps the first hidden layer's GEMM; the result must equal the serial path bit for bit."""
    _run_two_ranks(tmp_path, alg, port, mode, "gloo")


@pytest.mark.parametrize("alg,port,mode", [("FHADP", 29721, "serial"), ("INFADP", 29722, "serial"), ("FHADP", 29723, "overlap")])
def test_two_ranks_over_rccl(tmp_path, alg, port, mode):
    """The same check over RCCL (backend "nccl"), one GPU per rank - the path `bench.py --gpus N` and the sync trainers take on a
    multi-GPU node.  Needs two visible GPUs: skipped on the 1-GPU test box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip(f"{torch.cuda.device_count()} GPU visible: the RCCL path needs one GPU per rank")
    out = _run_two_ranks(tmp_path, alg, port, mode, "nccl")
    assert "(nccl, cuda:1)" in out


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (the way the driver runs `--gpus 1`): bench.py starts the two
    ranks itself; on this 1-GPU box they share the device and reduce over gloo, and the JSON line says so."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                          "--workload", "cfg1_idp_fhadp_b64_h10"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and rec["scaling"] == "weak"
    assert rec["value"] > 0 and rec["steps"] == 4
    import torch
    assert ("gloo" in rec["backend"]) == (torch.cuda.device_count() < 2)
    # the collective is instrumented: who reduced over what, how long one all-reduce of the gradient payload
    # takes on its own, and the step time with the overlap on and off
    assert rec["rccl_ranks"] == (0 if "gloo" in rec["backend"] else 2) and rec["gpus_visible"] == torch.cuda.device_count()
    assert max(rec["allreduce_payload_bytes"].values()) > 0 and rec["allreduce_us"]["value"] > 0
    assert all(rec["overlap"][k]["value"] > 0 and rec["overlap"][k]["ms_per_step"] > 0 for k in ("on", "off"))
