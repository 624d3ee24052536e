"""Premise check for a grouped (deferred) weight-gradient launch: R same-shape problems of a coarse level in ONE grid
(emulated with the existing kernel by replicating the pair lists R times: K' = R*27 'offsets') vs R separate launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from unscene3d_amd import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
def table(n, extent):
    c = rng.integers(-extent, extent, size=(4 * n, 3)); c[:, 2] = rng.integers(-2, 3, size=4 * n)
    c = np.unique(c, axis=0)[:n]
    c4 = np.concatenate([np.zeros((len(c), 1), np.int64), c], 1).astype(np.int32)
    cmap, _, _ = ops.coordmap_build(torch.from_numpy(c4).to(dev))
    return cmap, ops.kernel_map_cube(cmap, 3)
def timeit(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for n, extent, c, R in [(507, 9, 256, 11), (2222, 19, 128, 7), (2222, 19, 256, 3), (9402, 40, 64, 5), (9402, 40, 128, 3)]:
    cmap, nbr = table(n, extent)
    n = cmap.n
    rb = ops.rulebook_compact(nbr)
    P = rb.P
    a = torch.randn(n, c, device=dev); b = torch.randn(n, c, device=dev)
    K = 27
    dW = torch.zeros(K, c, c, device=dev)
    t1 = timeit(lambda: ops.wgrad(a, b, K, rb.in_idx, rb.out_idx, rb.koff, into=dW))
    # replicated lists
    in_r = rb.in_idx[:P].repeat(R); out_r = rb.out_idx[:P].repeat(R)
    koff = rb.koff.cpu().numpy()
    koffR = np.concatenate([koff[:-1] + r * P for r in range(R)] + [[R * P]]).astype(np.int64)
    koffR = torch.from_numpy(koffR).to(dev)
    dWR = torch.zeros(K * R, c, c, device=dev)
    tR = timeit(lambda: ops.wgrad(a, b, K * R, in_r.contiguous(), out_r.contiguous(), koffR, into=dWR))
    ok = torch.allclose(dWR[:K] , dWR[K:2*K]) if R > 1 else True
    print(f"n={n:5d} c={c:3d} P={P:6d}: single launch {t1:6.1f} us (x{R} = {t1*R:7.1f}); grouped x{R} in one launch {tR:7.1f} us = {tR/R:5.1f} per problem; consistent={ok}", flush=True)
