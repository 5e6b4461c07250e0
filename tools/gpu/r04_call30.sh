#!/bin/bash
# round 4, call 30: full GPU suite, smoke, default bench (what the driver runs), then the per-workload profiles (final kernels of the round)
root=$(pwd); out=$root/gpurun_out/r04_call30; mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $out/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $out/smoke.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 400 $out/bench_default.json; echo
TAG=r04 bash tools/collect_all_profiles.sh 2>&1 | tail -10
