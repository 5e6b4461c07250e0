#!/usr/bin/env python
"""Turn a gpurun_out/<tag>_<workload>[_f16]/ directory (bench JSON, rocprofv3 --stats CSV, one --pmc pass per
counter group; made by tools/collect_profile.sh on the GPU box) into the committed summary under profiles/:
    python tools/summarize_profile.py gpurun_out/r02_target_veh3dof_fhadp_b4096_h30
writes profiles/<dirname>_{bench.json, kernel_stats.csv, pmc_per_launch.json, summary.md}.
Every counter in the output comes from THIS collection (nothing is carried over from older profiles)."""
import collections
import csv
import json
import os
import shutil
import sys

src = sys.argv[1].rstrip("/")
name = os.path.basename(src)
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, f"{name}_kernel_stats.csv"))
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, f"{name}_bench.json"), "w"), indent=1)
updates = int(open(os.path.join(src, "updates.txt")).read()) if os.path.exists(os.path.join(src, "updates.txt")) else 1

KEEP = ("rollout", "dw_gemm", "dw_out", "reduce_partials", "adam", "prologue", "upload_params")
pmc = {}
for d in sorted(os.listdir(src)):
    f = os.path.join(src, d, "counter_collection.csv")
    if not d.startswith("pmc_") or not os.path.exists(f):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if any(x in k for x in KEEP):
            rec = pmc.setdefault(k, {})
            rec.update({c: sum(x) / len(x) for c, x in v.items()})
            rec["launches_per_update"] = max(len(x) for x in v.values()) / updates
json.dump(pmc, open(os.path.join(dst, f"{name}_pmc_per_launch.json"), "w"), indent=1)

stats = {r["Name"].split("(")[0].strip('"'): r for r in csv.DictReader(open(os.path.join(dst, f"{name}_kernel_stats.csv")))}
cb = bench.get("cpu_baseline", {})
rf = bench["roofline"]
lines = [f"# Profile summary {name} (MI355X, workload {bench['config']['workload']}, dtype {bench['dtype']})", "",
         f"bench.py: {bench['value'] / 1e6:.1f} M env-model steps/s, {bench['ms_per_step']:.3f} ms/step; roofline ({rf['bound']}, "
         f"{rf['kernel']}): {rf['achieved']:.1f} of {rf['peak']:.0f} {rf['unit']} = {rf['frac']:.3f}; whole update: "
         f"flops_fraction {bench.get('flops_fraction', float('nan')):.3f}, alg_hbm_fraction {bench.get('alg_hbm_fraction', float('nan')):.3f}; "
         f"CPU baseline ({cb.get('kind', '-')}, {cb.get('cores', '-')} threads): {cb.get('value', 0) / 1e3:.1f} k steps/s", "",
         "Per-launch averages: duration from `rocprofv3 --kernel-trace --stats`, counters from separate `--pmc` passes of this "
         "collection (FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B; FETCH/WRITE_SIZE are KiB).", "",
         "| kernel | launches / update | avg us (rocprof) | MFMA busy % | HBM read MB (corrected) | HBM write MB | HBM TB/s | L2 hit % | waves parked % |",
         "|---|---|---|---|---|---|---|---|---|"]
total_bytes = 0.0
for k, c in sorted(pmc.items(), key=lambda kv: -float(stats.get(kv[0], {}).get("TotalDurationNs", 0) or 0)):
    st = stats.get(k)
    avg_us = float(st["AverageNs"]) / 1e3 if st else float("nan")
    gui = c.get("GRBM_GUI_ACTIVE", 0) / 8.0   # summed over 8 XCDs
    mfma = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * gui) if gui else float("nan")
    rd = 2.0 * c["FETCH_SIZE"] * 1024 / 1e6 if "FETCH_SIZE" in c else float("nan")
    wr = c["WRITE_SIZE"] * 1024 / 1e6 if "WRITE_SIZE" in c else float("nan")
    if rd == rd and wr == wr:
        total_bytes += (rd + wr) * 1e6 * c.get("launches_per_update", 1.0)
    bw = (rd + wr) / avg_us if avg_us == avg_us and avg_us > 0 else float("nan")   # MB / us = TB/s
    hit = 100.0 * c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0))
    park = 100.0 * c.get("SQ_WAIT_ANY", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 0))
    lines.append(f"| `{k}` | {c.get('launches_per_update', 0):.2f} | {avg_us:.1f} | {mfma:.1f} | {rd:.1f} | {wr:.1f} | {bw:.2f} | {hit:.0f} | {park:.0f} |")
ms = bench["ms_per_step"]
lines += ["", f"HBM bytes of one update (all kernels above, counters): {total_bytes / 1e6:.0f} MB -> "
          f"{total_bytes / (ms * 1e-3) / 1e12:.2f} TB/s = hbm_fraction {total_bytes / (ms * 1e-3) / 8e12:.3f} of 8 TB/s "
          f"(algorithmic, SURVEY 8d: {bench.get('alg_hbm_fraction', float('nan')):.3f}).", "",
          "bench.py HIP-event timings of the same kernels (ms): " +
          ", ".join(f"{k}: {v['avg_ms']:.3f}" for k, v in bench["kernels_ms"].items()), ""]
open(os.path.join(dst, f"{name}_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines))
