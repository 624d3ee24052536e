#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection CSVs: mean counter value per dispatch, per kernel."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat and pat not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k[:110])
    for c, v in sorted(cs.items()):
        print(f"   {c:<36} mean/dispatch {sum(v)/len(v):16.1f}   n={len(v)}")
