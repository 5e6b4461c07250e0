#!/bin/bash
# A/B of library variants on one bench workload:
#   WL="--workload cfg5_lq_infadp_b65536 [--dtype fp16]" tools/gpu/ab_workload.sh <tag> <variant> [<variant> ...]   ("base" = the product library)
root=$(pwd); tag=$1; shift; out=$root/gpurun_out/$tag; mkdir -p $out
B="python bench.py $WL --no-other-workloads --no-cpu-baseline --steps ${STEPS:-20} --warmup 5"
i=0
for v in "$@"; do
    i=$((i + 1)); lib=gops_amd/libgops_hip_$v.so; [ "$v" = base ] && lib=gops_amd/libgops_hip.so
    GOPS_HIP_LIB=$lib timeout 600 $B > $out/bench_${i}_$v.json 2> $out/bench_${i}_$v.err
done
python - "$out" "$@" <<'PY'
import json, sys
out = sys.argv[1]
for i, n in enumerate(sys.argv[2:], 1):
    try:
        d = json.loads(open(f"{out}/bench_{i}_{n}.json").read().strip().splitlines()[-1])
        k = d["kernels_ms"]
        print(f"{n:12s} {d['value'] / 1e6:7.1f} M  " + "  ".join(f"{kk.split(' ')[0][:18]} {vv['avg_ms']:.3f}" for kk, vv in k.items()))
    except Exception as e:
        print(n, "failed", e, open(f"{out}/bench_{i}_{n}.err").read()[-500:])
PY
