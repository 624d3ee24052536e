#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory into a short text table."""
import csv
import glob
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {f}\n# total kernel time {tot/1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} launches")
print(f"{'kernel':<100} {'calls':>6} {'total_ms':>9} {'avg_us':>9} {'pct':>6}")
for r in rows[:top]:
    print(f"{r['Name'][:100]:<100} {r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:9.3f} {float(r['AverageNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}")
