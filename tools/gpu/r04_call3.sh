#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r04_call3; mkdir -p $out
export GOPS_SS_VEH=1
GOPS_HIP_LIB=gops_amd/libgops_hip_dump.so GOPS_SSB=0 timeout 600 python tools/gpu/dbg_dump_diff.py veh_p10 4 2>&1 | tee $out/dump.log | grep "sn0 tile\|sn4 tile" | cut -c1-400 | head -60
