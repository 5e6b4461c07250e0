#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call30; mkdir -p $out
timeout 900 python -m pytest tests/test_split_gpu.py -q -m gpu -s -k "streamed_split" 2>&1 | grep -E "rel-L2|passed|failed|Error|assert|error" | tail -20 | tee $out/pytest.log
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $out/pytest_all.log
