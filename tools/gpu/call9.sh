#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r03_call9; mkdir -p $out
echo "== full pytest"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $out/pytest_all.log
echo "== bench default"; timeout 900 python bench.py > $out/bench_all.json 2> $out/bench_all.err; python - <<PY
import json
d=json.load(open('$out/bench_all.json'))
print('target', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels_ms'].items()}, d['roofline']['frac'], d['kernel_variant'])
for k,v in d['workloads'].items(): print(k, round(v['value']/1e6,1), round(v['ms_per_step'],3), round(v['roofline']['frac'],3), {a:round(b,3) for a,b in v['kernels_ms'].items()})
print(d['cpu_baseline'])
PY
tail -3 $out/bench_all.err
