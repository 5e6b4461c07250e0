#!/bin/bash
# round 4, call 7: 64-row half-precision kernels - f16 parity suite + cfg5 fp16 bench
root=$(pwd); out=$root/gpurun_out/r04_call7; mkdir -p $out
timeout 1200 python -m pytest tests/test_f16_gpu.py -q -x 2>&1 | tail -15 | tee $out/pytest_f16.log
timeout 600 python bench.py --workload cfg5_lq_infadp_b65536 --dtype fp16 --no-other-workloads --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_cfg5_f16.json 2> $out/bench_cfg5_f16.err; tail -c 2500 $out/bench_cfg5_f16.json; tail -3 $out/bench_cfg5_f16.err
