#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call5; mkdir -p $out
b() { python -c "
import json,sys
d=json.load(open('$1')); print('$2', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels_ms'].items()})"; }
GOPS_DW_EXACT=1 python tools/dw_compare.py save /tmp/exact.pt 2>/dev/null
for v in "" _dwh2; do
  echo "== lib$v: dW vs exact"; GOPS_HIP_LIB=gops_amd/libgops_hip$v.so python tools/dw_compare.py save /tmp/h2$v.pt 2>/dev/null; python tools/dw_compare.py diff /tmp/h2$v.pt /tmp/exact.pt | tee $out/dw_diff$v.txt
  echo "== lib$v: parity"; GOPS_HIP_LIB=gops_amd/libgops_hip$v.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_split_gpu.py -q -m gpu -k "baseline_shapes or split" 2>&1 | tail -4
  for wl in target_veh3dof_fhadp_b4096_h30 cfg2_idp_fhadp_b4096_h30 cfg4_veh3dof_fhadp_b4096_h50; do
    GOPS_HIP_LIB=gops_amd/libgops_hip$v.so timeout 300 python bench.py --workload $wl --steps 60 --warmup 10 --no-cpu-baseline > $out/bench_${wl}$v.json 2> $out/bench_${wl}$v.err; b $out/bench_${wl}$v.json "$wl lib$v"
  done
done
echo "== exact for reference"; GOPS_DW_EXACT=1 timeout 300 python bench.py --workload target_veh3dof_fhadp_b4096_h30 --steps 60 --warmup 10 --no-cpu-baseline > $out/bench_exact.json 2>/dev/null; b $out/bench_exact.json "target exact"
echo "== full pytest (default lib)"; timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $out/pytest_all.log
