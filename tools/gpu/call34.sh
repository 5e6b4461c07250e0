#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call34; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; export GOPS_HIP_GRAPH=0
for v in 0 4; do
  export GOPS_TOUCH=$v
  rm -rf /tmp/pmc_$v
  timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_$v -o p -- python $root/tools/dbg_run.py cfg5_lq_infadp_b65536 fp32 6 > /tmp/pmc_$v.log 2>&1
  f=$(find /tmp/pmc_$v -name "*counter_collection.csv" | head -1)
  python - "$f" $v <<'PY' | tee -a $out/fetch.log
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] == 'FETCH_SIZE': acc[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k, v in acc.items():
    if 'rollout' in k: print('touch', sys.argv[2], k, len(v), 'avg MB (x2 corrected)', round(2 * sum(v) / len(v) * 1024 / 1e6, 1))
PY
done
