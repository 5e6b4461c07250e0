#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call23; mkdir -p $out
for i in 1 2 3; do
  for v in new pin6; do
    if [ $v = new ]; then unset GOPS_HIP_LIB; else export GOPS_HIP_LIB=$root/gops_amd/libgops_hip_$v.so; fi
    timeout 300 python bench.py --no-other-workloads --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x['avg_ms'],4) for k,x in d['kernels_ms'].items()})" | tee -a $out/ab.log
  done
done
for v in new pin6; do
  if [ $v = new ]; then unset GOPS_HIP_LIB; else export GOPS_HIP_LIB=$root/gops_amd/libgops_hip_$v.so; fi
  timeout 300 python bench.py --workload cfg2_idp_fhadp_b4096_h30 --no-other-workloads --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 $v', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x['avg_ms'],4) for k,x in d['kernels_ms'].items()})" | tee -a $out/ab.log
done
