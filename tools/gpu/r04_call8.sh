#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r04_call8; mkdir -p $out
timeout 1500 python -m pytest tests/test_multi_rank_gpu.py tests/test_alg_gpu.py -q -x 2>&1 | tail -12 | tee $out/pytest.log
timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --no-other-workloads --no-cpu-baseline > $out/bench_2rank.json 2> $out/bench_2rank.err; tail -c 600 $out/bench_2rank.json; tail -3 $out/bench_2rank.err
