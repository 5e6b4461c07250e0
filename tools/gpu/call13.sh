#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call13; mkdir -p $out
timeout 900 python -m pytest tests/test_alg_gpu.py -x -q -m gpu -k "opt_controller or get_constraint" 2>&1 | tail -60 | tee $out/pytest_optc.log
timeout 900 python -m pytest tests/test_mobilerobot_gpu.py tests/test_hip_parity.py -q -m gpu -k "mobilerobot or constrained or surr or spil" 2>&1 | tail -8 | tee $out/pytest_mob.log
