#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call21; mkdir -p $out
GOPS_HIP_LIB=$root/gops_amd/libgops_hip_dbg.so GOPS_DBG_TIMING=1 timeout 300 python tools/dbg_run.py cfg4_veh3dof_fhadp_b4096_h50 fp32 2>&1 | grep "gops dbg" | tail -2 | tee $out/dbg.log
GOPS_SPLIT_STREAM0=0 GOPS_HIP_LIB=$root/gops_amd/libgops_hip_dbg.so GOPS_DBG_TIMING=1 timeout 300 python tools/dbg_run.py cfg4_veh3dof_fhadp_b4096_h50 fp32 2>&1 | grep "gops dbg" | tail -2 | tee -a $out/dbg.log
