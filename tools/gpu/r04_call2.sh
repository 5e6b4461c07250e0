#!/bin/bash
# round 4, call 2: packed-fp32 hazard hypothesis - library built without v_pk_*_f32, and the micro-reproducer
root=$(pwd); out=$root/gpurun_out/r04_call2; mkdir -p $out
export GOPS_SS_VEH=1
echo "== micro reproducer"; timeout 300 tools/microbench/pk_hazard 2>&1 | tee $out/pk_hazard.txt
echo "== nopk fwd"; GOPS_HIP_LIB=gops_amd/libgops_hip_nopk.so GOPS_SSB=0 DBG_NOGRAD=1 timeout 600 python tools/gpu/dbg_poison.py veh_p10 veh_fhadp_3x256 2>&1 | tee $out/nopk_fwd.log | cut -c1-300 | tail -24
echo "== nopk sweep lq_many veh_p10"; GOPS_HIP_LIB=gops_amd/libgops_hip_nopk.so timeout 600 python tools/gpu/dbg_poison.py lq_many veh_p10 idp_many 2>&1 | tee $out/nopk_sweep.log | cut -c1-300 | tail -34
