"""Achieved HBM bandwidth of the memory-bound kernels at the bench scene's sizes:
algorithmic bytes per launch / measured device time, against the 8 TB/s peak and the 6.29 TB/s
the guide gives as achievable (MI355X_MICROARCH.md).

Every measured launch works on a DIFFERENT copy of its operands: the copies of one kernel add up to > 512 MB, twice
the 256 MB Infinity Cache, so a launch finds none of its input in MALL or L2 and the rate is an HBM rate (round 1
replayed 20 launches on one 57 MB tensor and reported cache rates).  Device time = HIP-graph replay of the rotation,
divided by the launches in it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unscene3d_amd import ops
from unscene3d_amd._lib import check, lib
from unscene3d_amd.synthetic import make_scene

PEAK, ACHIEVABLE = 8000.0, 6290.0
FOOTPRINT = 600e6
dev = torch.device("cuda:0")


def copies(nbytes):
    return max(2, int(FOOTPRINT // max(nbytes, 1)) + 1)


def timed(fns, graph=True, reps=3):
    """fns: one closure per operand copy; -> seconds per launch."""
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if graph:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for f in fns:
                    f()
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best / len(fns) * 1e-3
    e0.record()
    for f in fns:
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / len(fns) * 1e-3


rows = []


def rep(name, bytes_, sec, note=""):
    gbs = bytes_ / sec / 1e9
    rows.append(f"{name:50s} {bytes_/1e6:8.1f} MB {sec*1e6:8.1f} us {gbs:7.0f} GB/s {100*gbs/PEAK:5.1f}% of 8 TB/s "
                f"{100*gbs/ACHIEVABLE:5.1f}% of 6.29  {note}")


sc = make_scene(2000, target_voxels=150000)
xyz = torch.from_numpy(sc["xyz"]).to(dev)
P = xyz.shape[0]
R = copies(P * 36)
xs = [xyz.clone() for _ in range(R)]
rep("voxel_floor (f64 xyz -> i32)", P * (24 + 12), timed([lambda x=x: ops.voxel_floor(x, 0.02) for x in xs]))
del xs
ec = ops.voxel_floor(xyz, 0.02)
c4 = torch.cat([torch.zeros((P, 1), dtype=torch.int32, device=dev), ec], 1).contiguous()
rep("coordmap_build (points -> unique voxels)", P * (16 + 8 + 8),
    timed([lambda: ops.coordmap_build(c4)] * 10, graph=False), "incl. its count read-back (latency)")
cmap, uidx, inv = ops.coordmap_build(c4)
N = cmap.n
rep("kernel_map_cube k3 (27 probes / voxel)", N * 16 + 27 * N * 16 + 27 * N * 4,
    timed([lambda: ops.kernel_map_cube(cmap, 3)] * 10), "16 B per probe counted; table is cache resident by design")
nbr = ops.kernel_map_cube(cmap, 3)
coarse, _, parent = ops.coordmap_build(cmap.coords, quant=2, tensor_stride=2)
nbr2, kidx = ops.kernel_map_down2(cmap, parent, coarse)

stream = ops._stream
for n, C in ((N, 96), (N, 64), (N, 32), (coarse.n, 96), (9402, 128), (2222, 256), (507, 256)):
    R = copies(4 * n * C)
    x = [torch.randn(n, C, device=dev) for _ in range(R)]
    dy = [torch.randn(n, C, device=dev) for _ in range(R)]
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    st = torch.empty(4, C, device=dev)
    red = torch.empty(4, C, device=dev)
    ws = torch.empty(lib.usc_colstats_ws_bytes(n, C), dtype=torch.uint8, device=dev)
    out = torch.empty(n, C, device=dev)

    def f_stats(i):
        check(lib.usc_bn_forward_stats(x[i].data_ptr(), n, C, gam.data_ptr(), bet.data_ptr(), 1e-5, 0.1, None, None, None,
                                       st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(),
                                       ws.data_ptr(), ws.numel(), stream()))

    def f_apply(i):
        check(lib.usc_bn_apply(x[i].data_ptr(), st[2].data_ptr(), st[3].data_ptr(), dy[i].data_ptr(), 1, out.data_ptr(),
                               n, C, stream()))

    def f_red(i):
        check(lib.usc_bn_backward_reduce(x[i].data_ptr(), dy[i].data_ptr(), x[(i + 1) % R].data_ptr(), st[0].data_ptr(),
                                         st[1].data_ptr(), n, C, 1, 0, red[0].data_ptr(), red[1].data_ptr(),
                                         red[2].data_ptr(), red[3].data_ptr(), ws.data_ptr(), ws.numel(), stream()))

    def f_dx(i):
        check(lib.usc_bn_backward_dx(x[i].data_ptr(), dy[i].data_ptr(), x[(i + 1) % R].data_ptr(), st[0].data_ptr(),
                                     st[1].data_ptr(), gam.data_ptr(), red[2].data_ptr(), red[3].data_ptr(),
                                     out.data_ptr(), None, n, C, stream()))

    f_stats(0); torch.cuda.synchronize()
    tag = f"n={n} C={C}"
    rep(f"BN statistics + finalise           {tag}", 4 * n * C, timed([lambda i=i: f_stats(i) for i in range(R)]))
    rep(f"BN apply + residual + ReLU         {tag}", 12 * n * C, timed([lambda i=i: f_apply(i) for i in range(R)]))
    rep(f"BN backward reduce (x, dy, y)      {tag}", 12 * n * C, timed([lambda i=i: f_red(i) for i in range(R)]))
    rep(f"BN backward dx (x, dy, y -> dx)    {tag}", 16 * n * C, timed([lambda i=i: f_dx(i) for i in range(R)]))
    if n == N:
        rep(f"ReLU fwd                           {tag}", 8 * n * C, timed([lambda i=i: ops.relu(x[i]) for i in range(R)]))
        idx = torch.randint(0, n, (n,), device=dev)
        rep(f"gather_rows (random rows)          {tag}", 8 * n * C + 8 * n,
            timed([lambda i=i: ops.gather_rows(x[i], idx) for i in range(R)]))
        # scatter-add of row gradients (the backward of the gather: dst[idx[i]] += src[i]) through a permutation, i.e.
        # without duplicates, like the decoder's sampled keys: src read + dst read-modify-write + indices
        perm = torch.randperm(n, device=dev)
        dst = [torch.zeros(n, C, device=dev) for _ in range(R)]

        def f_scatter(i):
            check(lib.usc_scatter_add_rows(x[i].data_ptr(), C, perm.data_ptr(), n, dst[i].data_ptr(), stream()))
        rep(f"scatter_add_rows (atomics)         {tag}", 12 * n * C + 8 * n, timed([lambda i=i: f_scatter(i) for i in range(R)]))

        def f_scatter_u(i):
            check(lib.usc_scatter_rows_unique(x[i].data_ptr(), C, perm.data_ptr(), n, dst[i].data_ptr(), stream()))
        rep(f"scatter_rows_unique (permutation)  {tag}", 8 * n * C + 8 * n, timed([lambda i=i: f_scatter_u(i) for i in range(R)]))
        del dst
    del x, dy
R = copies(400 * N)
m100 = [torch.randn(N, 100, device=dev) for _ in range(R)]
rep("avgpool k2s2 on [N,100] mask logits", 4 * 100 * (N + coarse.n) + 8 * N,
    timed([lambda m=m: ops.avgpool_down2(m, nbr2) for m in m100]))
del m100
seg = torch.randint(0, 1500, (N,), device=dev)
csr = ops.segment_csr(seg, 1500)
R = copies(512 * N)
f128 = [torch.randn(N, 128, device=dev) for _ in range(R)]
rep("segment_csr (stable counting sort)", 8 * N * 3, timed([lambda: ops.segment_csr(seg, 1500)] * 10), "latency-bound (4 launches)")
rep("segment_mean fwd d=128", 4 * 128 * N + 8 * N + 4 * 128 * 1500, timed([lambda f=f: ops.segment_mean(f, csr) for f in f128]))
del f128
pts = xyz[uidx.long()].float()[None].contiguous()
rep("furthest_point_sample m=100", 100 * N * 12, timed([lambda: ops.furthest_point_sample(pts, 100)] * 5),
    "algorithmic re-read; the points are register resident: latency-bound by 100 dependent rounds")
print(f"# {N} voxels from {P} points; peak {PEAK:.0f} GB/s, achievable {ACHIEVABLE:.0f} GB/s; every launch of a rotation "
      f"reads a different copy of its operands (> {FOOTPRINT/1e6:.0f} MB per rotation)")
print("\n".join(rows))
