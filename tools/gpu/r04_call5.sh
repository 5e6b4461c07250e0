#!/bin/bash
# round 4, call 5: full GPU suite + default bench with the library built without packed-fp32 instructions
root=$(pwd); out=$root/gpurun_out/r04_call5; mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 | tee $out/pytest_all.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 3000 $out/bench_default.json; tail -5 $out/bench_default.err
