#!/usr/bin/env python
"""Timeline of ONE update from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py / tools/dbg_run.py:
    python tools/timeline.py <dir with *kernel_trace.csv> [anchor kernel substring = prologue_kernel]
prints every kernel between the last two launches of the anchor kernel: start offset, duration and the idle gap in front
of it (us), then the sums - where an update's time goes besides the three big kernels."""
import csv
import glob
import os
import sys

d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "prologue_kernel"
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
if len(idx) < 3:
    sys.exit(f"fewer than three launches of {anchor}")
a, b = idx[-3], idx[-2]   # a complete update in the middle of the run
t0 = rows[a][0]
busy = gaps = 0.0
prev_end = rows[a - 1][1] if a > 0 else t0
print(f"{'start us':>9} {'dur us':>8} {'gap us':>7}  kernel")
for s, e, name in rows[a:b]:
    gap = (s - prev_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {name[:110]}")
    busy += (e - s) / 1e3
    gaps += max(gap, 0.0)
    prev_end = max(prev_end, e)
print(f"update: {(rows[b][0] - t0) / 1e3:.1f} us start to start; kernels {busy:.1f} us, idle gaps {gaps:.1f} us ({len(rows[a:b])} launches)")
