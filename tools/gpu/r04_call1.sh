#!/bin/bash
# round 4, call 1: reproducibility hunt - baseline, forced-zero waitcnt build, per-thread dump build
root=$(pwd); out=$root/gpurun_out/r04_call1; mkdir -p $out
export GOPS_SS_VEH=1
echo "== baseline fwd (SSB=0, no grad)"; GOPS_SSB=0 DBG_NOGRAD=1 timeout 600 python tools/gpu/dbg_poison.py veh_p10 2>&1 | tee $out/base_fwd.log | cut -c1-400 | tail -40
echo "== forcezero fwd"; GOPS_HIP_LIB=gops_amd/libgops_hip_fz.so GOPS_SSB=0 DBG_NOGRAD=1 timeout 600 python tools/gpu/dbg_poison.py veh_p10 2>&1 | tee $out/fz_fwd.log | cut -c1-400 | tail -40
echo "== dump build"; GOPS_HIP_LIB=gops_amd/libgops_hip_dump.so GOPS_SSB=0 timeout 600 python tools/gpu/dbg_dump_diff.py veh_p10 8 2>&1 | tee $out/dump.log | cut -c1-600 | tail -80
echo "== baseline sweep lq_many"; timeout 600 python tools/gpu/dbg_poison.py lq_many 2>&1 | tee $out/base_sweep.log | cut -c1-300 | tail -14
echo "== forcezero sweep lq_many"; GOPS_HIP_LIB=gops_amd/libgops_hip_fz.so timeout 600 python tools/gpu/dbg_poison.py lq_many 2>&1 | tee $out/fz_sweep.log | cut -c1-300 | tail -14
