#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call19; mkdir -p $out
timeout 600 python -m pytest tests/test_split_gpu.py tests/test_hip_parity.py -q -m gpu -k "split or veh or stationary or baseline" 2>&1 | tail -3 | tee $out/pytest.log
GOPS_HIP_LIB=$root/gops_amd/libgops_hip_dbg.so GOPS_DBG_TIMING=1 timeout 300 python tools/dbg_run.py target_veh3dof_fhadp_b4096_h30 fp32 2>&1 | grep "gops dbg" | tail -2 | tee $out/dbg.log
for i in 1 2 3; do
  for v in new noenvp; do
    if [ $v = new ]; then unset GOPS_HIP_LIB; else export GOPS_HIP_LIB=$root/gops_amd/libgops_hip_$v.so; fi
    timeout 300 python bench.py --no-other-workloads --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x['avg_ms'],4) for k,x in d['kernels_ms'].items()})" | tee -a $out/ab.log
  done
done
