"""Developer aid: can the fine-level weight gradients run BESIDE the latency-bound coarse-level chain of the backward
pass?  Stream A: the input-gradient convolutions + batch-norm launches of the 9 402 / 2 222 / 507-row levels (what the
backward pass does between block7 and block1); stream B: the weight gradients of the 148 564-row level, launched
normally or in the background form (usc_spconv_wgrad_grid_limit).  Alone, alone, together."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unscene3d_amd import MinkowskiEngine as ME, ops
from unscene3d_amd._lib import lib
from unscene3d_amd.synthetic import make_scene

dev = torch.device("cuda:0")
sc = make_scene(2000, target_voxels=150000)
c3, umap, _ = ME.utils.sparse_quantize(sc["xyz"], quantization_size=0.02, return_index=True, return_inverse=True, device="cuda:0")
coords = torch.cat([torch.zeros((c3.shape[0], 1), dtype=torch.int32, device=dev), c3], 1).contiguous()
coords = ops.gather_rows_i32(coords, ops.spatial_order(coords))
x = ME.SparseTensor(features=torch.zeros(coords.shape[0], 3, device=dev), coordinates=coords, device=dev)
cm = x.coordinate_manager
for ts in (1, 2, 4, 8):
    cm.stride_map(ts)

chain = []
for ts, cin, reps in [(4, 128, 6), (4, 64, 10), (8, 256, 6), (8, 128, 14), (16, 256, 22), (8, 128, 8), (4, 64, 6)]:
    n = cm.coord_map(ts).n
    nbr = cm.cube_map(ts)["nbr"]
    xin = torch.randn(n, cin, device=dev)
    W = torch.randn(27, cin, cin, device=dev) * 0.05
    g, b = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    chain.append((n, nbr, xin, W, g, b, reps))


def run_chain():
    for n, nbr, xin, W, g, b, reps in chain:
        y = xin
        for _ in range(reps):
            y = ops.gather_gemm(y, W, nbr, n, w_transposed=True)
            y = ops.batch_norm_act(y, g, b, None, True, 1e-5, None, None, 0.02, True)


n1 = cm.coord_map(1).n
rb = cm.cube_rulebook(1)
a96, b96, a128 = torch.randn(n1, 96, device=dev), torch.randn(n1, 96, device=dev), torch.randn(n1, 128, device=dev)
dW = [torch.zeros(27, 96, 96, device=dev) for _ in range(3)] + [torch.zeros(27, 128, 96, device=dev)]


def run_wgrads():
    for j in range(3):
        ops.wgrad(a96, b96, 27, rb.in_idx, rb.out_idx, rb.koff, into=dW[j])
    ops.wgrad(a128, b96, 27, rb.in_idx, rb.out_idx, rb.koff, into=dW[3])


sA, sB = torch.cuda.Stream(), torch.cuda.Stream(priority=0)
sBlow = torch.cuda.Stream(priority=0)


def wall(fnA, fnB, reps=5):
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        e0, eA, eB = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        sA.wait_event(e0); sB.wait_event(e0)
        if fnB is not None:
            with torch.cuda.stream(sB):
                fnB()
                eB.record(sB)
        if fnA is not None:
            with torch.cuda.stream(sA):
                fnA()
                eA.record(sA)
        torch.cuda.synchronize()
        ts.append((e0.elapsed_time(eA) if fnA is not None else 0.0, e0.elapsed_time(eB) if fnB is not None else 0.0))
    ts = ts[1:]
    return sum(t[0] for t in ts) / len(ts), sum(t[1] for t in ts) / len(ts)


run_chain(); run_wgrads(); torch.cuda.synchronize()
print(f"chain alone       : {wall(run_chain, None)[0]:.3f} ms")
for lim in (0, 512, 256, 128, 64):
    lib.usc_spconv_wgrad_grid_limit(lim)
    wa = wall(None, run_wgrads)[1]
    ca, cb = wall(run_chain, run_wgrads)
    print(f"grid limit {lim:4d}: wgrads alone {wa:.3f} ms | together: chain done {ca:.3f} ms, wgrads done {cb:.3f} ms")
lib.usc_spconv_wgrad_grid_limit(0)
