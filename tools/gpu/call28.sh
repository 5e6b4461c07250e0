#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call28; mkdir -p $out
timeout 900 python -m pytest tests/test_split_gpu.py tests/test_hip_parity.py -q -m gpu -k "split or baseline or stationary or idp" 2>&1 | tail -4 | tee $out/pytest.log
for v in 0 1; do
  if [ $v = 1 ]; then export GOPS_NO_FUSED_DWOUT=1; else unset GOPS_NO_FUSED_DWOUT; fi
  timeout 300 python bench.py --workload cfg2_idp_fhadp_b4096_h30 --no-other-workloads --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 nofuse=$v', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x['avg_ms'],4) for k,x in d['kernels_ms'].items()})" | tee -a $out/cfg2.log
done
