#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call4; mkdir -p $out
echo "== pytest split"; timeout 900 python -m pytest tests/test_split_gpu.py -x -q -s -m gpu 2>&1 | grep -v "^$" | tail -22 | tee $out/pytest_split.log
echo "== full pytest"; timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $out/pytest_all.log
echo "== bench default"; timeout 900 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $out/bench_all.json 2> $out/bench_all.err; python - <<PY
import json
d=json.load(open('$out/bench_all.json'))
print('target', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels_ms'].items()})
for k,v in d['workloads'].items(): print(k, round(v['value']/1e6,1), round(v['ms_per_step'],3), {a:round(b,3) for a,b in v['kernels_ms'].items()})
PY
echo "== kernel trace target"; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/stats; GOPS_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats -o s -- python $root/bench.py --workload target_veh3dof_fhadp_b4096_h30 --steps 40 --warmup 10 --no-cpu-baseline > /tmp/stats.log 2>&1; cp $(find /tmp/stats -name "*kernel_stats.csv" | head -1) $out/target_kernel_stats.csv; head -25 $out/target_kernel_stats.csv | cut -c1-150
