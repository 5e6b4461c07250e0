#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call14; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee $out/pytest_all.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 1500 $out/bench_default.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
