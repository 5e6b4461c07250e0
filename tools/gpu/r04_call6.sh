#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r04_call6; mkdir -p $out
for lib in gops_amd/libgops_hip.so ; do
echo "== $lib"; GOPS_HIP_LIB=$lib timeout 600 python -m pytest tests/test_hip_parity.py -q -x -k "test_env_step_vs_reference_fixture" 2>&1 | grep -E "AssertionError: \(|passed|failed" | cut -c1-200
done
