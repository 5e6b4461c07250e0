#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call25; mkdir -p $out
for v in 0 1; do
  if [ $v = 1 ]; then export GOPS_SPLIT_TAIL_MULTI=1; else unset GOPS_SPLIT_TAIL_MULTI; fi
  timeout 300 python bench.py --workload cfg5_lq_infadp_b65536 --no-other-workloads --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 tailmulti=$v', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x['avg_ms'],4) for k,x in d['kernels_ms'].items()})" | tee -a $out/ab.log
done
unset GOPS_SPLIT_TAIL_MULTI
timeout 300 python bench.py --workload cfg5_lq_infadp_b65536 --dtype fp16 --no-other-workloads --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 f16', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x['avg_ms'],4) for k,x in d['kernels_ms'].items()})" | tee -a $out/ab.log
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_split_gpu.py tests/test_f16_gpu.py -q -m gpu -k "lq or cartpole or pendulum or cfg5 or tail" 2>&1 | tail -3 | tee -a $out/ab.log
