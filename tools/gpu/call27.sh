#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call27; mkdir -p $out
timeout 900 python -m pytest tests/test_split_gpu.py tests/test_hip_parity.py tests/test_alg_gpu.py -q -m gpu -k "split or baseline or gemm or sweep or stationary" 2>&1 | tail -4 | tee $out/pytest.log
timeout 300 python bench.py --workload cfg4_veh3dof_fhadp_b4096_h50 --no-other-workloads --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x['avg_ms'],4) for k,x in d['kernels_ms'].items()})" | tee -a $out/cfg4.log
