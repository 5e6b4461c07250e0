#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call35; mkdir -p $out
timeout 900 python -m pytest tests/test_split_gpu.py tests/test_hip_parity.py -q -m gpu -x 2>&1 | tail -8 | tee $out/pytest.log
run() {  # label
  for w in target_veh3dof_fhadp_b4096_h30 cfg5_lq_infadp_b65536 cfg3_veh3dof_infadp_b8192 cfg4_veh3dof_fhadp_b4096_h50 cfg2_idp_fhadp_b4096_h30; do
  timeout 300 python bench.py --workload $w --no-other-workloads --no-cpu-baseline --steps 60 --warmup 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $1', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x['avg_ms'],4) for k,x in d['kernels_ms'].items() if 'value' not in k})" | tee -a $out/ab.log
  done
}
run spec
GOPS_DW_SPEC=0 run ring
