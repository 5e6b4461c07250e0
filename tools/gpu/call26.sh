#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call26; mkdir -p $out
for i in 1 2 3; do
  for v in auto 1; do
    export GOPS_HIP_GRAPH=$v
    timeout 300 python bench.py --no-other-workloads --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph=$v', round(d['value']/1e6,2), round(d['ms_per_step'],4))" | tee -a $out/ab.log
  done
done
