#!/usr/bin/env python
"""Sections of one step from a stream_timeline.py listing: per queue busy time, the compute queue's sections (backbone
forward / decoder forward / criterion / decoder backward / backbone backward by U-Net level), and what the other queues
ran during each.  Usage: python tools/timeline_sections.py <timeline_last_step.txt>"""
import re
import sys
from collections import defaultdict

rows = []
for ln in open(sys.argv[1]):
    m = re.match(r"\s*([\d.]+)\s+([\d.]+) gap\s+(-?[\d.]+)\s+q(\d+)\s+(.*)", ln)
    if m:
        rows.append((float(m.group(1)), float(m.group(2)), int(m.group(4)), m.group(5).strip()))
busy = defaultdict(float)
cnt = defaultdict(int)
for t, d, q, n in rows:
    busy[q] += d
    cnt[q] += 1
main_q = max(cnt, key=lambda q: cnt[q])
end = max(t + d for t, d, q, n in rows)
print(f"step span {end / 1e3:.2f} ms; compute queue = q{main_q}")
for q in sorted(busy):
    ks = [r for r in rows if r[2] == q]
    print(f"  q{q}: {cnt[q]:5d} kernels, busy {busy[q] / 1e3:6.2f} ms, first at {ks[0][0] / 1e3:6.2f}, last ends {max(r[0] + r[1] for r in ks) / 1e3:6.2f} ms")
mk = [r for r in rows if r[2] == main_q]
def first(name, after=0.0):
    for t, d, q, n in mk:
        if name in n and t >= after:
            return t
    return None
marks = [("backbone fwd", first("stem_conv_kernel")), ("decoder fwd", first("fps_multi_kernel")), ("criterion", first("crit_")),
         ("decoder bwd", first("crit_bwd")), ]
# backbone backward: first conv-ish kernel after the last attention backward kernel
last_attn = max([t + d for t, d, q, n in mk if "attn" in n or "layernorm_bwd" in n] or [0.0])
marks.append(("backbone bwd", last_attn))
marks.append(("optimizer", first("adamw_kernel")))
marks = [(a, b) for a, b in marks if b is not None]
marks.sort(key=lambda x: x[1])
print("sections of the compute queue:")
for i, (name, t0) in enumerate(marks):
    t1 = marks[i + 1][1] if i + 1 < len(marks) else end
    ks = [r for r in mk if t0 <= r[0] < t1]
    b = sum(r[1] for r in ks)
    other = {q: sum(min(r[0] + r[1], t1) - max(r[0], t0) for r in rows if r[2] == q and r[0] < t1 and r[0] + r[1] > t0) for q in busy if q != main_q}
    print(f"  {name:14s} {t0 / 1e3:6.2f} -> {t1 / 1e3:6.2f} ms ({(t1 - t0) / 1e3:5.2f}): {len(ks):4d} kernels busy {b / 1e3:5.2f} ms, idle {(t1 - t0 - b) / 1e3:5.2f}; "
          + ", ".join(f"q{q} {v / 1e3:.2f}" for q, v in sorted(other.items()) if v > 1))
# backbone backward by row count is not in the trace; list its big kernels with what ran beside them
t0 = dict(marks).get("backbone bwd")
if t0 is not None:
    print("backbone backward, compute-queue kernels >= 100 us and concurrent other-queue kernels >= 100 us:")
    for t, d, q, n in mk:
        if t >= t0 and d >= 100:
            con = [f"q{r[2]}:{r[3][:28]}({r[1]:.0f})" for r in rows if r[2] != main_q and r[1] >= 100 and r[0] < t + d and r[0] + r[1] > t]
            print(f"   {t / 1e3:6.2f} {d:6.0f} {n[:44]:44s} | {' '.join(con)}")
