#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call6; mkdir -p $out
echo "== pytest split"; timeout 900 python -m pytest tests/test_split_gpu.py -x -q -s -m gpu 2>&1 | grep -v "^$" | tail -8 | tee $out/pytest_split.log
echo "== full pytest"; timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $out/pytest_all.log
echo "== dbg cycles (split)"; GOPS_HIP_LIB=gops_amd/libgops_hip_dbg.so GOPS_DBG_TIMING=1 GOPS_HIP_GRAPH=0 timeout 300 python tools/dbg_run.py target_veh3dof_fhadp_b4096_h30 fp32 3 2>&1 | grep "gops dbg" | tail -2 | tee $out/dbg_split.log
echo "== bench default"; timeout 900 python bench.py > $out/bench_all.json 2> $out/bench_all.err; python - <<PY
import json
d=json.load(open('$out/bench_all.json'))
print('target', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels_ms'].items()})
for k,v in d['workloads'].items(): print(k, round(v['value']/1e6,1), round(v['ms_per_step'],3), {a:round(b,3) for a,b in v['kernels_ms'].items()})
PY
