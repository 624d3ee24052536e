"""TEST INFRASTRUCTURE ONLY — CPU (numpy, fp32) restatement of the reference's furthest point sampling.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(unscene3d_amd/pointnet2_utils.py -> usc_furthest_point_sample, csrc/points.hip) never does.

PARITY UNPINNED: the reference has this path only as a CUDA extension
(third_party/pointnet2/_ext_src/src/sampling_gpu.cu) without a test or a golden vector, and it cannot run in a
container without an NVIDIA device.  The restatement is anchored on the source text:

* running minimum distance per point, start index 0        sampling_gpu.cu:73-115 (per-thread strided scan, `d2 > best` keeps
  the FIRST maximum a thread meets)
* points with |p|^2 <= 1e-3 never update and never win      sampling_gpu.cu:103-104
* block-wide argmax by halving strides (__update)          sampling_gpu.cu:62-71, 118-171 — with ties the entry at
  the LOWER thread slot survives a halving step, so among equal distances the winner is the candidate with the
  smallest (index mod block size), then the smallest index; block size = largest power of two <= n, capped at 512
  (sampling_gpu.cu:176-216, opt_n_threads)
"""
from __future__ import annotations

import numpy as np


def furthest_point_sample(xyz, m):
    """xyz [n, 3] float32 -> m int32 indices, bit-exact with the reference kernel's tie-break (block size <= 512)."""
    n = xyz.shape[0]
    bs = 1
    while bs * 2 <= n and bs < 512:
        bs *= 2
    tmp = np.full(n, 1e10, np.float32)
    idx = np.zeros(m, np.int32)
    mag = (xyz[:, 0] * xyz[:, 0] + xyz[:, 1] * xyz[:, 1] + xyz[:, 2] * xyz[:, 2]).astype(np.float32)
    ok = ~(mag <= np.float32(1e-3))
    ks = np.arange(n)
    old = 0
    for j in range(1, m):
        d = ((xyz - xyz[old]) ** 2).astype(np.float32)
        d = (d[:, 0] + d[:, 1] + d[:, 2]).astype(np.float32)
        d2 = np.minimum(d, tmp)
        tmp = np.where(ok, d2, tmp)
        cand = np.where(ok, d2, -np.inf)
        best = cand.max()
        if not np.isfinite(best):
            old = 0
        else:
            tied = ks[cand == best]
            old = int(tied[np.lexsort((tied, tied % bs))][0])
        idx[j] = old
    return idx
