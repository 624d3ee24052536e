#!/usr/bin/env python
"""CPU time per Python thread and step: how much interpreter work do the main thread (the step) and the prefetcher's worker
(voxelisation + maps of the scene after the next) do, against the step's wall time?  The two share the interpreter lock:
if their CPU times add up to the wall time, the step is bound by the interpreter, not by the device.
    python tools/host_threads.py [bench flags]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

args = bench.parse()
dev = torch.device("cuda:0")
step = bench.make_mask3d_step(args, dev, 0, 1)
worker = {"cpu": 0.0, "wall": 0.0, "n": 0}
if step.prefetch is not None:
    orig = step.prefetch._issue

    def issue(*a, **k):
        c0, w0 = time.thread_time(), time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            worker["cpu"] += time.thread_time() - c0
            worker["wall"] += time.perf_counter() - w0
            worker["n"] += 1
    step.prefetch._issue = issue
for w in range(args.warmup):
    step(1)
    if w == 1 and os.environ.get("USC3D_STEADY", "1") == "1":
        from unscene3d_amd.trainer.trainer import prepare_steady_state
        prepare_steady_state(dev)
torch.cuda.synchronize()
worker.update(cpu=0.0, wall=0.0, n=0)
c0, w0 = time.thread_time(), time.perf_counter()
for _ in range(args.steps):
    step(1)
c1, w1 = time.thread_time(), time.perf_counter()
torch.cuda.synchronize()
w2 = time.perf_counter()
n = args.steps
print(f"per step over {n} steps: wall {1e3 * (w2 - w0) / n:.2f} ms (host loop returned after {1e3 * (w1 - w0) / n:.2f}); "
      f"main thread CPU {1e3 * (c1 - c0) / n:.2f} ms; worker CPU {1e3 * worker['cpu'] / max(1, worker['n']):.2f} ms "
      f"(wall inside _issue {1e3 * worker['wall'] / max(1, worker['n']):.2f} ms, {worker['n']} batches)")
step.close()
