#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call12; mkdir -p $out
timeout 900 python -m pytest tests/test_mobilerobot_gpu.py -q -m gpu 2>&1 | tail -60 | tee $out/pytest_mob.log
