#!/usr/bin/env python
"""cProfile of the full training step (host side) — finds Python/launch overhead hot spots."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

args = bench.parse()
# backward on the calling thread so that cProfile sees the Python backward of the custom Functions as well
torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda:0")
step = bench.make_mask3d_step(args, dev, 0, 1)
for _ in range(2):
    step(1)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
import time  # noqa: E402
t0 = time.perf_counter()
for _ in range(3):
    step(1)
torch.cuda.synchronize()
pr.disable()
print(f"wall {1e3 * (time.perf_counter() - t0) / 3:.1f} ms per step under the profiler")
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
