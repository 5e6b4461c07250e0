#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call7; mkdir -p $out
b() { python -c "
import json,sys
d=json.load(open('$1')); print('$2', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels_ms'].items()})"; }
for g in 0 1; do
  GOPS_HIP_GRAPH=$g timeout 300 python bench.py --workload target_veh3dof_fhadp_b4096_h30 --steps 100 --warmup 20 --no-cpu-baseline > $out/bench_graph$g.json 2> $out/bench_graph$g.err; b $out/bench_graph$g.json "target graph=$g"
  GOPS_HIP_GRAPH=$g timeout 300 python bench.py --workload cfg2_idp_fhadp_b4096_h30 --steps 100 --warmup 20 --no-cpu-baseline > $out/bench_cfg2_graph$g.json 2> /dev/null; b $out/bench_cfg2_graph$g.json "cfg2 graph=$g"
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tl
GOPS_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $root/bench.py --workload target_veh3dof_fhadp_b4096_h30 --steps 12 --warmup 4 --no-cpu-baseline > /tmp/tl.log 2>&1
python $root/tools/timeline.py /tmp/tl | tee $out/timeline_target_eager.txt
