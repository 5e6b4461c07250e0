#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call35; mkdir -p $out
timeout 1200 python -m pytest tests/test_split_gpu.py tests/test_hip_parity.py tests/test_alg_gpu.py -q -m gpu -x 2>&1 | tail -8 | tee $out/pytest.log
run() {  # label
  for w in cfg5_lq_infadp_b65536 cfg2_idp_fhadp_b4096_h30; do
  timeout 300 python bench.py --workload $w --no-other-workloads --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $1', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x['avg_ms'],4) for k,x in d['kernels_ms'].items()})" | tee -a $out/ab.log
  done
}
run skinny
GOPS_DW_SKINNY=0 run fm
