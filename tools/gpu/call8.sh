#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call8; mkdir -p $out
echo "== full pytest"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $out/pytest_all.log
b() { python -c "
import json,sys
d=json.load(open('$1')); print('$2', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels_ms'].items()})"; }
for wl in target_veh3dof_fhadp_b4096_h30 cfg2_idp_fhadp_b4096_h30; do
  timeout 300 python bench.py --workload $wl --steps 100 --warmup 20 --no-cpu-baseline > $out/bench_$wl.json 2> $out/bench_$wl.err; b $out/bench_$wl.json "$wl"
  GOPS_NO_FUSED_DWOUT=1 timeout 300 python bench.py --workload $wl --steps 100 --warmup 20 --no-cpu-baseline > $out/bench_${wl}_nofuse.json 2> /dev/null; b $out/bench_${wl}_nofuse.json "$wl no fused dw_out"
  GOPS_EAGER_LOG=1 timeout 300 python bench.py --workload $wl --steps 100 --warmup 20 --no-cpu-baseline > $out/bench_${wl}_eagerlog.json 2> /dev/null; b $out/bench_${wl}_eagerlog.json "$wl eager log"
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tl
GOPS_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $root/bench.py --workload target_veh3dof_fhadp_b4096_h30 --steps 12 --warmup 4 --no-cpu-baseline > /tmp/tl.log 2>&1
python $root/tools/timeline.py /tmp/tl | tee $out/timeline_target_eager.txt
