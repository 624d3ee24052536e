#!/usr/bin/env python
"""Where the device idles: gaps between kernels in a rocprofv3 --kernel-trace CSV (all streams merged).
Usage: python tools/idle_gaps.py <dir with *_kernel_trace.csv> [steps=14] [min_gap_us=20]
Caveat: the tracer slows the host (bench step 30 -> 43 ms), so the gaps show where the HOST is late under tracing —
useful to rank host-side stalls (the assignment read-back, the prefetch hand-over), not to size them."""
import csv
import glob
import sys
from collections import Counter

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 14
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the steady state: the last `steps` adamw launches delimit the steps
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
if len(ad) > 4:
    lo, hi = ad[2], ad[-1]          # skip the first two (warm-up)
    n_steps = len(ad) - 3
    rows = rows[lo:hi + 1]
else:
    n_steps = steps
busy_end = rows[0][1]
gaps = []
prev = rows[0][2]
for s, e, name in rows[1:]:
    if s > busy_end:
        gaps.append(((s - busy_end) / 1e3, prev, name))
    if e > busy_end:
        busy_end, prev = e, name
span = (rows[-1][1] - rows[0][0]) / 1e6
idle = sum(g[0] for g in gaps) / 1e3
print(f"{n_steps} steps, {span / n_steps:.2f} ms per step, device idle {idle / n_steps:.2f} ms per step in {len(gaps) / n_steps:.0f} gaps")
big = [g for g in gaps if g[0] >= min_gap]
print(f"gaps >= {min_gap:.0f} us: {sum(g[0] for g in big) / 1e3 / n_steps:.2f} ms per step in {len(big) / n_steps:.1f} gaps")
c = Counter()
t = Counter()
for g, a, b in big:
    key = (a[:60], b[:60])
    c[key] += 1
    t[key] += g
for key, tot in t.most_common(15):
    print(f"  {tot / 1e3 / n_steps:6.3f} ms/step  {c[key] / n_steps:4.1f}x  after {key[0]}  ->  {key[1]}")
small = [g[0] for g in gaps if g[0] < min_gap]
print(f"gaps < {min_gap:.0f} us: {sum(small) / 1e3 / n_steps:.2f} ms per step, mean {sum(small) / max(1, len(small)):.2f} us")
