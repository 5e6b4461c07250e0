#!/bin/bash
# Run ON THE GPU BOX: per-kernel durations (rocprofv3 --kernel-trace --stats) of a few updates of one workload:
#     bash tools/kstats.sh [workload] [fp32|fp16] [updates]      -> prints the top of the stats table
wl=${1:-target_veh3dof_fhadp_b4096_h30}; dt=${2:-fp32}; n=${3:-20}
root=$(pwd); cd /tmp && export TMPDIR=/tmp; export GOPS_HIP_GRAPH=0
rm -rf /tmp/kst
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o s -- python $root/tools/dbg_run.py $wl $dt $n > /tmp/kst.log 2>&1
f=$(find /tmp/kst -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}  {r['Percentage']}%")
PY
