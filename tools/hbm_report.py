"""Achieved HBM bandwidth of the memory-bound kernels on the bench scene's stride-1 map:
algorithmic bytes per launch / measured duration, against the 8 TB/s peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unscene3d_amd import MinkowskiEngine as ME, ops
from unscene3d_amd.synthetic import make_scene

PEAK = 8000.0
dev = torch.device("cuda:0")
def t(fn, n=20, graph=True):
    """Device time per call: n calls captured in one HIP graph (no host launch overhead in the measurement);
    graph=False for ops that read a count back to the host."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if graph:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for _ in range(n): fn()
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
sc = make_scene(2000, target_voxels=150000)
xyz = torch.from_numpy(sc["xyz"]).to(dev)
rows = []
def rep(name, bytes_, sec, note=""):
    rows.append(f"{name:44s} {bytes_/1e6:9.1f} MB {sec*1e6:9.1f} us {bytes_/sec/1e9:8.0f} GB/s  {100*bytes_/sec/1e9/PEAK:5.1f}% of peak  {note}")
P = xyz.shape[0]
ec = ops.voxel_floor(xyz, 0.02)
rep("voxel_floor (f64 xyz -> i32)", P * (24 + 12), t(lambda: ops.voxel_floor(xyz, 0.02)))
c4 = torch.cat([torch.zeros((P, 1), dtype=torch.int32, device=dev), ec], 1).contiguous()
rep("coordmap_build (points -> unique voxels)", P * (16 + 8 + 8), t(lambda: ops.coordmap_build(c4), graph=False), "incl. count read-back")
cmap, uidx, inv = ops.coordmap_build(c4)
N = cmap.n
rep("kernel_map_cube k3 (27 probes / voxel)", N * 16 + 27 * N * 16 + 27 * N * 4, t(lambda: ops.kernel_map_cube(cmap, 3)), "16 B per probe counted")
nbr = ops.kernel_map_cube(cmap, 3)
rep("rowsort_build (masks + bucket sort)", 27 * N * 4 + N * 12, t(lambda: (setattr(nbr, "_usc_rowsort", None), ops.rowsort(nbr))))
coarse, _, parent = ops.coordmap_build(cmap.coords, quant=2, tensor_stride=2)
nbr2, kidx = ops.kernel_map_down2(cmap, parent, coarse)
for C in (96, 32):
    x = torch.randn(N, C, device=dev); dy = torch.randn(N, C, device=dev); res = torch.randn(N, C, device=dev)
    g = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev, requires_grad=True)
    rep(f"BN stats (colstats) C={C}", 4 * N * C, t(lambda: ops.colstats(x)))
    sc_, sh_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rep(f"BN apply + residual + ReLU C={C}", 12 * N * C, t(lambda: ops.bn_apply(x, sc_, sh_, residual=res, relu=True)))
    xr = x.clone().requires_grad_()
    y = ops.batch_norm_act(xr, g, b, None, True)
    rep(f"BN backward (reduce + dx) C={C}", (12 + 16) * N * C, t(lambda: torch.autograd.grad(y, xr, dy, retain_graph=True), graph=False), "3 reads + 3 reads/1 write; eager loop: host-bound, see rocprof for kernel times")
    rep(f"ReLU fwd C={C}", 8 * N * C, t(lambda: ops.relu(x)))
    idx = torch.randint(0, N, (N,), device=dev)
    rep(f"gather_rows (random) C={C}", 8 * N * C + 8 * N, t(lambda: ops.gather_rows(x, idx)))
m100 = torch.randn(N, 100, device=dev)
rep("avgpool k2s2 on [N,100] mask logits", 4 * 100 * (N + coarse.n) + 8 * N, t(lambda: ops.avgpool_down2(m100, nbr2)))
seg = torch.randint(0, 1500, (N,), device=dev)
csr = ops.segment_csr(seg, 1500)
f128 = torch.randn(N, 128, device=dev)
rep("segment_csr (stable counting sort)", 8 * N * 3, t(lambda: ops.segment_csr(seg, 1500)))
rep("segment_mean fwd d=128", 4 * 128 * N + 8 * N + 4 * 128 * 1500, t(lambda: ops.segment_mean(f128, csr)))
pts = xyz[uidx.long()].float()[None].contiguous()
rep("furthest_point_sample m=100", 100 * N * 12, t(lambda: ops.furthest_point_sample(pts, 100), n=5), "algorithmic re-read; points are register resident")
print(f"# {N} voxels from {P} points; peak {PEAK:.0f} GB/s")
print("\n".join(rows))
