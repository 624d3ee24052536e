#!/usr/bin/env python
"""Developer aid: is the training step bound by the host (Python + launch calls) or by the device?
Per step, starting from an idle device: the time until step() returns (everything issued) and the time until the
device is done.  issue ~ done: the host is the bottleneck (the device finishes right behind the last launch);
issue << done: the device is.  Pipelined steady state (bench.py) = max of the two, roughly.
    python tools/host_vs_device.py [bench flags]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

args = bench.parse()
dev = torch.device("cuda:0")
step = bench.make_mask3d_step(args, dev, 0, 1)
for _ in range(5):
    step(1)
torch.cuda.synchronize()
issue, done = [], []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    issue.append(1e3 * (t1 - t0))
    done.append(1e3 * (t2 - t0))
issue.sort()
done.sort()
# free-running for comparison
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step(1)
torch.cuda.synchronize()
free = 1e3 * (time.perf_counter() - t0) / 20
print(f"issued after {issue[len(issue) // 2]:.2f} ms (median), device done after {done[len(done) // 2]:.2f} ms; "
      f"free-running {free:.2f} ms per step")
