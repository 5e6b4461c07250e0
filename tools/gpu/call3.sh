#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call3; mkdir -p $out
echo "== full pytest"; timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $out/pytest_all.log
echo "== dbg cycles (split)"; GOPS_HIP_LIB=gops_amd/libgops_hip_dbg.so GOPS_DBG_TIMING=1 GOPS_HIP_GRAPH=0 timeout 300 python tools/dbg_run.py target_veh3dof_fhadp_b4096_h30 fp32 3 2>&1 | grep "gops dbg" | tail -2 | tee $out/dbg_split.log
b() { python -c "
import json,sys
d=json.load(open('$1')); print('$2', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels_ms'].items()})"; }
for v in "" _pin1 _pin2; do
  for wl in target_veh3dof_fhadp_b4096_h30 cfg2_idp_fhadp_b4096_h30; do
    GOPS_HIP_LIB=gops_amd/libgops_hip$v.so timeout 300 python bench.py --workload $wl --steps 60 --warmup 10 --no-cpu-baseline > $out/bench_${wl}$v.json 2> $out/bench_${wl}$v.err; b $out/bench_${wl}$v.json "$wl lib$v"
  done
done
