#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:
#     bash tools/collect_profile.sh r02 target_veh3dof_fhadp_b4096_h30 [fp32|fp16]
# Writes gpurun_out/<tag>_<workload>[_f16]/{bench.json, kernel_stats.csv, pmc_<i>/counter_collection.csv};
# tools/summarize_profile.py then turns that directory into profiles/<tag>_<workload>[_f16]_*.
# Every counter group is collected in its OWN short run (kernel-trace only): FETCH_SIZE and WRITE_SIZE do not fit
# the TCC's 4 slots together (3 + 2, MI355X_MICROARCH.md), which is what made the r01 pass time out.
tag=${1:-r03}
wl=${2:-target_veh3dof_fhadp_b4096_h30}
dt=${3:-fp32}
sfx=""; [ "$dt" = "fp16" ] && sfx="_f16"
root=$(pwd)
out=$root/gpurun_out/${tag}_${wl}${sfx}
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export GOPS_HIP_GRAPH=0
# the algorithm classes' PrecisionGuard re-forms the first gradient (and every 500th) on the exact-fp32 kernels: 1 update in 8 of these
# short runs would carry those launches - per-update averages of the DEFAULT kernels are wanted here
export GOPS_PRECISION_CHECK_INTERVAL=0
UPDATES=8
timeout 400 python $root/bench.py --workload $wl --dtype $dt --steps 100 --warmup 20 > $out/bench.json 2> $out/bench.err
rm -rf /tmp/stats
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats -o s -- python $root/bench.py --workload $wl --dtype $dt --steps 40 --warmup 10 --no-cpu-baseline > /tmp/stats.log 2>&1
cp $(find /tmp/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
echo $UPDATES > $out/updates.txt
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_$i -o p -- python $root/tools/dbg_run.py $wl $dt $UPDATES > /tmp/pmc_$i.log 2>&1
  echo "pmc pass $i ($ctrs) rc=$?"
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then mkdir -p $out/pmc_$i; cp $f $out/pmc_$i/counter_collection.csv; else tail -5 /tmp/pmc_$i.log; fi
done
ls -R $out | head -30
