#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call32; mkdir -p $out
timeout 900 python -m pytest tests/test_split_gpu.py -q -m gpu -s -k "streamed_split" 2>&1 | grep -E "rel-L2|passed|failed|Error|assert|error" | tail -20 | tee $out/pytest.log
for v in 1 0; do
  export GOPS_SSB=$v
  for w in cfg5_lq_infadp_b65536; do
  timeout 300 python bench.py --workload $w --no-other-workloads --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w ssb=$v', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x['avg_ms'],4) for k,x in d['kernels_ms'].items() if 'value' not in k})" | tee -a $out/ab.log
  done
done
