#!/bin/bash
# GPU call 1 of round 3: evidence + baselines before the kernel work.
root=$(pwd); out=$root/gpurun_out/r03_call1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
echo "== microbench: mfma_valu_overlap" ; timeout 120 $root/tools/microbench/ovl > $out/mfma_valu_overlap.txt 2>&1; tail -3 $out/mfma_valu_overlap.txt
echo "== microbench: mfma_filler_curve" ; timeout 120 $root/tools/microbench/curve > $out/mfma_filler_curve.txt 2>&1; cat $out/mfma_filler_curve.txt
rm -rf /tmp/ovl_pmc
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /tmp/ovl_pmc -o p -- $root/tools/microbench/ovl > /tmp/ovl_pmc.log 2>&1
f=$(find /tmp/ovl_pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/mfma_valu_overlap_pmc.csv || tail -5 /tmp/ovl_pmc.log
cd $root
echo "== pytest (new tests)"
timeout 900 python -m pytest tests/test_multi_rank_gpu.py tests/test_alg_gpu.py -x -q -m gpu -k "two_ranks or bench_starts or schedule_advances or constrained_fhadp" 2>&1 | tail -15 | tee $out/pytest_new.log
echo "== bench default (all workloads)"
timeout 900 python bench.py --steps 50 --warmup 10 > $out/bench_baseline.json 2> $out/bench_baseline.err; tail -c 600 $out/bench_baseline.json; tail -3 $out/bench_baseline.err
echo "== bench 2 ranks (self-launched; gloo on one GPU)"
timeout 600 python bench.py --gpus 2 --steps 30 --warmup 10 --workload target_veh3dof_fhadp_b4096_h30 > $out/bench_2rank.json 2> $out/bench_2rank.err; tail -c 400 $out/bench_2rank.json; tail -3 $out/bench_2rank.err
