#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call33; mkdir -p $out
for v in 2 0 6 4; do
  export GOPS_TOUCH=$v
  for w in cfg5_lq_infadp_b65536 cfg3_veh3dof_infadp_b8192; do
  timeout 300 python bench.py --workload $w --no-other-workloads --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w touch=$v', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x['avg_ms'],4) for k,x in d['kernels_ms'].items() if 'value' not in k})" | tee -a $out/ab.log
  done
done
