#!/usr/bin/env python
"""Developer aid: the kernel sequence of ONE steady-state step from a rocprofv3 --kernel-trace CSV
(queue, start offset in us, duration in us, grid, workgroup, kernel), in launch order.
Usage: python tools/step_kernel_sequence.py <dir with *_kernel_trace.csv> > sequence.txt"""
import csv
import glob
import re
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"),
                     r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
rows.sort()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
lo, hi = ad[-3], ad[-2]      # the last TIMED step (bench.py appends one instrumented step for the roofline)
t0 = rows[lo + 1][0]
qs = {}
for s, e, n, q, g, w in rows[lo + 1:hi + 1]:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "at::", n)
    qi = qs.setdefault(q, len(qs))
    print(f"q{qi} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {g:>9s} {w:>5s}  {n[:150]}")
