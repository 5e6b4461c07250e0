#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call15; mkdir -p $out
echo "== GOPS_SPLIT=0" | tee $out/knobs.log
GOPS_SPLIT=0 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_alg_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee -a $out/knobs.log
echo "== GOPS_HIP_GRAPH=1" | tee -a $out/knobs.log
GOPS_HIP_GRAPH=1 timeout 900 python -m pytest tests/test_alg_gpu.py tests/test_mobilerobot_gpu.py tests/test_multi_rank_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee -a $out/knobs.log
echo "== GOPS_DW_EXACT=1 GOPS_EAGER_LOG=1" | tee -a $out/knobs.log
GOPS_DW_EXACT=1 GOPS_EAGER_LOG=1 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_split_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee -a $out/knobs.log
