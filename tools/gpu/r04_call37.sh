#!/bin/bash
# round 4, call 37: full GPU suite, smoke, default bench, profiles of the INFADP workloads (ABI v11 loss / polyak kernels)
root=$(pwd); out=$root/gpurun_out/r04_call37; mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $out/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $out/smoke.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 300 $out/bench_default.json; echo
for spec in "cfg3_veh3dof_infadp_b8192 fp32" "cfg5_lq_infadp_b65536 fp32" "cfg5_lq_infadp_b65536 fp16"; do
  set -- $spec
  bash tools/collect_profile.sh r04 $1 $2 2>&1 | grep -E "pmc pass|rror" | tr '\n' ' ' | cut -c1-200; echo " <- $1 $2"
done
