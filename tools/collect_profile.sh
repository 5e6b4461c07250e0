#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/collect_profile.sh r01
# Writes gpurun_out/<tag>/{bench_<tag>.json, stats/<tag>_kernel_stats.csv, pmc_*/p_counter_collection.csv};
# tools/summarize_profile.py then turns that directory into profiles/<tag>_*.
# Counters are collected in their own passes (kernel-trace only), each under a timeout.
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out/stats
cd /tmp && export TMPDIR=/tmp
export GOPS_HIP_GRAPH=0
timeout 300 python $root/bench.py --steps 200 --warmup 20 > $out/bench_$tag.json 2> $out/bench.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats -o s -- python $root/bench.py --steps 50 --warmup 10 --no-cpu-baseline > /tmp/stats.log 2>&1
cp $(find /tmp/stats -name "*kernel_stats.csv" | head -1) $out/stats/${tag}_kernel_stats.csv
cat > /tmp/short_run.py <<PY
import sys, contextlib, torch
sys.path.insert(0, "$root")
from bench import alg_kwargs
from gops_amd.create_pkg.create_alg import create_alg
from gops_amd.utils.synthetic import CONFIGS, make_batch
cfg = CONFIGS["target_veh3dof_fhadp_b4096_h30"]
with contextlib.redirect_stdout(sys.stderr):
    alg = create_alg(**alg_kwargs(cfg, 0))
alg.networks.to("cuda")
data = {k: v.cuda() for k, v in make_batch(cfg, 1000).items()}
for it in range(6):
    alg.local_update(data, it)
torch.cuda.synchronize()
PY
i=0
for ctrs in "FETCH_SIZE WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_$i -o p -- python /tmp/short_run.py > /tmp/pmc_$i.log 2>&1
  echo "pmc pass $i ($ctrs) rc=$?"
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then mkdir -p $out/pmc_$i; cp $f $out/pmc_$i/p_counter_collection.csv; fi
done
ls -R $out | head -30
