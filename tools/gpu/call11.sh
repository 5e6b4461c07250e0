#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call11; mkdir -p $out
timeout 900 python -m pytest tests/test_mobilerobot_gpu.py -x -q -m gpu 2>&1 | tail -40 | tee $out/pytest_mob.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_mobilerobot_gpu.py 2>&1 | tail -15 | tee $out/pytest_all.log
