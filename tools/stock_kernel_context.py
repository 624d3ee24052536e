#!/usr/bin/env python
"""Developer aid: for every stock (non-usc) kernel of ONE steady-state step in a rocprofv3 --kernel-trace CSV, the
kernels launched just before and after it on the same queue — enough to tell which part of the step issues the
copies / adds / fills that are left.
Usage: python tools/stock_kernel_context.py <dir with *_kernel_trace.csv> [top=60]"""
import csv
import glob
import re
import sys
from collections import Counter

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"at::native::(\w+)<.*?(CUDAFunctor_add|FillFunctor<\w+>|MinMaxOps|func_wrapper_t<\w+|\w+Functor\w*)", n)
    if m:
        return f"at::{m.group(1)}[{m.group(2)}]"
    return n.split("(")[0][:60]


ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
lo, hi = ad[-3], ad[-2]      # the last TIMED step (bench.py appends one instrumented step for the roofline)
step = rows[lo + 1:hi + 1]
stock = Counter()
ctx = Counter()
per_q = {}
for r in step:
    per_q.setdefault(r[3], []).append(r)
n_all = len(step)
for q, rs in per_q.items():
    for i, r in enumerate(rs):
        if "usc::" in r[2]:
            continue
        name = short(r[2])
        stock[name] += 1
        prev = short(rs[i - 1][2]) if i else "-"
        nxt = short(rs[i + 1][2]) if i + 1 < len(rs) else "-"
        ctx[(name, prev, nxt)] += 1
t_stock = sum((r[1] - r[0]) for r in step if "usc::" not in r[2]) / 1e3
print(f"# one step: {n_all} launches on {len(per_q)} queues; stock kernels {sum(stock.values())} launches, {t_stock:.1f} us")
for q, rs in sorted(per_q.items(), key=lambda kv: -len(kv[1])):
    st = [r for r in rs if "usc::" not in r[2]]
    print(f"#   queue {q}: {len(rs)} launches, {sum(r[1] - r[0] for r in rs) / 1e3:.1f} us of kernels; stock {len(st)} launches, "
          f"{sum(r[1] - r[0] for r in st) / 1e3:.1f} us: " + ", ".join(f"{v} {k[:40]}" for k, v in Counter(short(r[2]) for r in st).most_common(8)))
for k, v in stock.most_common(40):
    print(f"{v:5d}  {k}")
print("# stock kernel | previous | next")
for (n, p, x), v in ctx.most_common(top):
    print(f"{v:4d}  {n:44s} | {p:50s} | {x}")
