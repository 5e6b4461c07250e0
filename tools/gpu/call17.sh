#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call17; mkdir -p $out
for i in 1 2; do
  for v in 512 256 384 768 1024; do
    export GOPS_DW_WGS=$v
    timeout 300 python bench.py --no-other-workloads --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x['avg_ms'],4) for k,x in d['kernels_ms'].items()})" | tee -a $out/ab.log
  done
done
