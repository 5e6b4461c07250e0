#!/bin/bash
# GPU call 2: first run of the plane-split stationary kernels.
root=$(pwd); out=$root/gpurun_out/r03_call2; mkdir -p $out
echo "== pytest split"; timeout 600 python -m pytest tests/test_split_gpu.py -x -q -s -m gpu 2>&1 | tail -25 | tee $out/pytest_split.log
echo "== pytest baseline shapes"; timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -k "baseline_shapes" 2>&1 | tail -8 | tee $out/pytest_shapes.log
echo "== dbg cycles (split)"; GOPS_HIP_LIB=gops_amd/libgops_hip_dbg.so GOPS_DBG_TIMING=1 GOPS_HIP_GRAPH=0 timeout 300 python tools/dbg_run.py target_veh3dof_fhadp_b4096_h30 fp32 3 2>&1 | grep "gops dbg" | tail -4 | tee $out/dbg_split.log
echo "== dbg cycles (fp32 mfma)"; GOPS_SPLIT=0 GOPS_HIP_LIB=gops_amd/libgops_hip_dbg.so GOPS_DBG_TIMING=1 GOPS_HIP_GRAPH=0 timeout 300 python tools/dbg_run.py target_veh3dof_fhadp_b4096_h30 fp32 3 2>&1 | grep "gops dbg" | tail -2 | tee $out/dbg_fp32.log
for wl in target_veh3dof_fhadp_b4096_h30 cfg2_idp_fhadp_b4096_h30; do
  echo "== bench $wl split"; timeout 300 python bench.py --workload $wl --steps 50 --warmup 10 --no-cpu-baseline > $out/bench_${wl}_split.json 2> $out/bench_${wl}_split.err; python -c "
import json,sys
d=json.load(open('$out/bench_${wl}_split.json')); print(round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels_ms'].items()})"
  echo "== bench $wl fp32 mfma"; GOPS_SPLIT=0 timeout 300 python bench.py --workload $wl --steps 50 --warmup 10 --no-cpu-baseline > $out/bench_${wl}_fp32.json 2> $out/bench_${wl}_fp32.err; python -c "
import json,sys
d=json.load(open('$out/bench_${wl}_fp32.json')); print(round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels_ms'].items()})"
done
