#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call24; mkdir -p $out
for v in 0 1; do
  if [ $v = 1 ]; then export GOPS_SPLIT_TAIL_MULTI=1; else unset GOPS_SPLIT_TAIL_MULTI; fi
  for w in cfg5_lq_infadp_b65536 cfg3_veh3dof_infadp_b8192; do
  timeout 300 python bench.py --workload $w --no-other-workloads --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w tailmulti=$v', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x['avg_ms'],4) for k,x in d['kernels_ms'].items()})" | tee -a $out/ab.log
  done
done
GOPS_SPLIT_TAIL_MULTI=1 GOPS_HIP_LIB=$root/gops_amd/libgops_hip_dbg.so GOPS_DBG_TIMING=1 timeout 300 python tools/dbg_run.py cfg5_lq_infadp_b65536 fp32 2>&1 | grep "gops dbg" | tail -4 | tee -a $out/ab.log
