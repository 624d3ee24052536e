#!/usr/bin/env python
"""Per-queue timeline of a rocprofv3 --kernel-trace CSV: how much of the work on secondary HIP streams (prefetch, the
decoder's key-preparation stream) really runs BESIDE the compute stream.
Usage: python tools/stream_overlap.py <dir with *_kernel_trace.csv> [top=12]
Steps are delimited by adamw launches; the first two steps are skipped.  (The tracer slows the host: read the overlap
fractions, not the absolute step time.)"""
import csv
import glob
import sys
from collections import Counter, defaultdict

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
lo, hi = ad[2], ad[-1]
n_steps = len(ad) - 3
rows = rows[lo + 1:hi + 1]
byq = defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
main_q = max(byq, key=lambda q: sum(e - s for s, e, _, _ in byq[q]))
span = (rows[-1][1] - rows[0][0]) / 1e6 / n_steps
print(f"{n_steps} steps, {span:.2f} ms per step under the tracer; compute queue = {main_q}")


def union(iv):
    iv = sorted(iv)
    out, cs, ce = [], None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            out.append((cs, ce))
            cs, ce = s, e
    if cs is not None:
        out.append((cs, ce))
    return out


def overlap(a, b):          # total intersection of two sorted disjoint interval lists
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


main_u = union([(s, e) for s, e, _, _ in byq[main_q]])
all_u = union([(s, e) for s, e, _, _ in rows])
print(f"busy: compute queue {sum(e - s for s, e in main_u) / 1e6 / n_steps:.2f} ms, any queue {sum(e - s for s, e in all_u) / 1e6 / n_steps:.2f} ms per step")
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    u = union([(s, e) for s, e, _, _ in rs])
    busy = sum(e - s for s, e in u)
    ov = overlap(u, main_u) if q != main_q else 0
    print(f"queue {q}: {len(rs) / n_steps:7.1f} launches/step, busy {busy / 1e6 / n_steps:6.3f} ms/step, "
          f"of which beside a compute-queue kernel {ov / 1e6 / n_steps:6.3f} ms")
    if True:
        c, t = Counter(), Counter()
        for s, e, name, _ in rs:
            c[name[:70]] += 1
            t[name[:70]] += e - s
        for name, tot in t.most_common(top):
            print(f"      {tot / 1e6 / n_steps:6.3f} ms/step {c[name] / n_steps:6.1f}x  {name}")
# slowdown of compute-queue kernels while another queue is busy: mean duration by name, overlapped vs not
other_u = union([(s, e) for s, e, _, q in rows if q != main_q])
dur = defaultdict(lambda: [0, 0, 0, 0])
for s, e, name, _ in byq[main_q]:
    o = overlap([(s, e)], other_u)
    k = dur[name[:70]]
    if o > 0.5 * (e - s):
        k[0] += 1
        k[1] += e - s
    else:
        k[2] += 1
        k[3] += e - s
print("compute-queue kernels, mean us when overlapped by another queue vs alone (>= 20 samples each):")
lst = [(k[1] - k[0] * (k[3] / k[2]), name, k) for name, k in dur.items() if k[0] >= 20 and k[2] >= 20]
for extra, name, k in sorted(lst, reverse=True)[:top]:
    print(f"      {k[1] / k[0] / 1e3:7.2f} vs {k[3] / k[2] / 1e3:7.2f} us  ({k[0] / n_steps:.1f} of {(k[0] + k[2]) / n_steps:.1f} per step; {extra / 1e6 / n_steps:+.3f} ms/step)  {name}")
