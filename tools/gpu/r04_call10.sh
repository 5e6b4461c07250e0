#!/bin/bash
# round 4, call 10: A/B of the 64-row half sweep - layer step in two feature halves at three workgroups per CU (libgops_hip_halves.so)
root=$(pwd); out=$root/gpurun_out/r04_call10; mkdir -p $out
B="python bench.py --workload cfg5_lq_infadp_b65536 --dtype fp16 --no-other-workloads --no-cpu-baseline --steps 20 --warmup 5"
timeout 600 $B > $out/bench_base.json 2> $out/bench_base.err
GOPS_HIP_LIB=gops_amd/libgops_hip_halves.so timeout 600 $B > $out/bench_halves.json 2> $out/bench_halves.err
GOPS_HIP_LIB=gops_amd/libgops_hip_halves.so timeout 1200 python -m pytest tests/test_f16_gpu.py -q -x 2>&1 | tail -5 | tee $out/pytest_f16_halves.log
python - <<'PY'
import json
for n in ("base", "halves"):
    try:
        d = json.loads(open(f"gpurun_out/r04_call10/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d.get("kernels_ms"), d["roofline"])
    except Exception as e:
        print(n, "failed", e)
PY
