import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from unscene3d_amd import ops
from unscene3d_amd.synthetic import make_segment_scene
from unscene3d_amd.pseudo_masks import ncut as N
feats, conn, label = make_segment_scene(7)
dev = torch.device("cuda:0")
f = tuple(torch.from_numpy(x).to(dev) for x in feats)
A, D = N.get_affinity_matrix(f, tau=0.6)
for _ in range(3):
    N.second_smallest_eigenvector(A, D)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
    for _ in range(5):
        N.second_smallest_eigenvector(A, D)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total / 5) for e in prof.key_averages() if e.device_time_total > 0]
for k, c, t in sorted(rows, key=lambda r: -r[2])[:12]:
    print(f"{t:10.1f} us/call  x{c/5:6.1f}  {k[:110]}")
