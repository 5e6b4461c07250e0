#!/bin/bash
# round 4, call 42: full GPU suite, smoke, default bench, cfg3 profile (plane-split step loop + exact tail)
root=$(pwd); out=$root/gpurun_out/r04_call42; mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $out/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $out/smoke.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 300 $out/bench_default.json; echo
bash tools/collect_profile.sh r04 cfg3_veh3dof_infadp_b8192 fp32 2>&1 | grep -E "rror" | head -3
