#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call10; mkdir -p $out
timeout 900 python -m pytest tests/test_alg_gpu.py -x -q -m gpu -k "opt_controller" 2>&1 | tail -15 | tee $out/pytest_optc.log
