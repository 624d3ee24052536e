#!/usr/bin/env python
"""Merged kernel timeline (all queues) of the LAST step in a rocprofv3 --kernel-trace CSV, for reading cross-stream
stalls: one line per kernel — start (us from the step's first kernel), duration, gap to the previous kernel of the same
queue, queue, name.  Usage: python tools/stream_timeline.py <dir> [from_us] [to_us] [step_from_end]
step_from_end: 1 = the span between the last two AdamW launches (bench.py: the instrumented per-operator step that
follows the timed loop), 2 (default) = the one before it = the LAST TIMED step."""
import csv
import glob
import sys

d = sys.argv[1]
lo_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi_us = float(sys.argv[3]) if len(sys.argv) > 3 else 1e12
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
back = int(sys.argv[4]) if len(sys.argv) > 4 else 2
rows = rows[ad[-back - 1] + 1:ad[-back] + 1]
t0 = rows[0][0]
last_end = {}
for s, e, name, q in rows:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    t = (s - t0) / 1e3
    if lo_us <= t <= hi_us:
        short = name.replace("usc::(anonymous namespace)::", "usc::").replace("void ", "")[:70]
        print(f"{t:10.1f} {(e - s) / 1e3:8.1f} gap {gap:8.1f}  q{q}  {short}")
