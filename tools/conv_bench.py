#!/usr/bin/env python
"""Per-shape micro-benchmark of the sparse-conv kernels on the bench scene's real kernel maps.
Usage (GPU box): python tools/conv_bench.py [--voxels 150000] [--reps 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unscene3d_amd import MinkowskiEngine as ME  # noqa: E402
from unscene3d_amd import ops  # noqa: E402
from unscene3d_amd.synthetic import make_scene  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=150000)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--sorted", action="store_true", help="apply ops.spatial_order to the voxel rows first")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sc = make_scene(2000, target_voxels=a.voxels)
    c3, umap, _ = ME.utils.sparse_quantize(sc["xyz"], quantization_size=0.02, return_index=True, return_inverse=True,
                                           device="cuda:0")
    coords = torch.cat([torch.zeros((c3.shape[0], 1), dtype=torch.int32, device=dev), c3], 1).contiguous()
    if a.sorted:
        coords = ops.gather_rows_i32(coords, ops.spatial_order(coords))
    x = ME.SparseTensor(features=torch.zeros(coords.shape[0], 3, device=dev), coordinates=coords, device=dev)
    cm = x.coordinate_manager
    for ts in (1, 2, 4, 8):
        cm.stride_map(ts)
    shapes = {1: [(3, 32), (128, 96), (96, 96)], 2: [(32, 32), (128, 96), (96, 96)], 4: [(32, 64), (64, 64), (192, 128), (128, 128)],
              8: [(64, 128), (128, 128), (384, 256), (256, 256)], 16: [(128, 256), (256, 256)]}
    print(f"{'level':>5} {'N':>7} {'shape':>10} {'P':>9} | {'fwd ms':>8} {'TF':>6} | {'dgrad ms':>8} {'TF':>6} | {'wgrad ms':>8} {'TF':>6}")
    for ts, lst in shapes.items():
        n = cm.coord_map(ts).n
        nbr = cm.cube_map(ts)["nbr"]
        rb = cm.cube_rulebook(ts)
        for cin, cout in lst:
            if a.only and a.only != f"{ts}:{cin}x{cout}":
                continue
            xin = torch.randn(n, cin, device=dev)
            W = torch.randn(27, cin, cout, device=dev) * 0.05
            dy = torch.randn(n, cout, device=dev)
            fl = 2.0 * rb.P * cin * cout
            t_f = timeit(lambda: ops.gather_gemm(xin, W, nbr, n), a.reps)
            Wt = ops.weight_transpose(W, True)
            t_d = timeit(lambda: ops.gather_gemm(dy, Wt, nbr, n), a.reps) if cin != 3 else float("nan")
            t_w = timeit(lambda: ops.wgrad(xin, dy, 27, rb.in_idx, rb.out_idx, rb.koff), a.reps)
            print(f"{ts:>5} {n:>7} {cin:>4}x{cout:<5} {rb.P:>9} | {t_f*1e3:8.3f} {fl/t_f/1e12:6.1f} | {t_d*1e3:8.3f} {fl/t_d/1e12:6.1f} | {t_w*1e3:8.3f} {fl/t_w/1e12:6.1f}")
    # strided / transposed
    print("k2s2 down / up (pairs form)")
    for ts, (cin, cout) in {1: (32, 32), 2: (32, 32), 4: (64, 64), 8: (128, 128)}.items():
        d = cm.stride_map(ts)
        rb = cm.down_rulebook(ts)
        nf, nc = cm.coord_map(ts).n, cm.coord_map(2 * ts).n
        xin = torch.randn(nf, cin, device=dev)
        W = torch.randn(8, cin, cout, device=dev) * 0.05
        fl = 2.0 * rb.P * cin * cout
        t_f = timeit(lambda: ops.gather_gemm(xin, W, d["nbr2"], nc), a.reps)
        dyc = torch.randn(nc, cout, device=dev)
        Wt = ops.weight_transpose(W, False)
        t_d = timeit(lambda: ops.pairs_gemm(dyc, Wt, rb.out_idx, rb.in_idx, rb.koff, rb.P, nf), a.reps)
        t_w = timeit(lambda: ops.wgrad(xin, dyc, 8, rb.in_idx, rb.out_idx, rb.koff), a.reps)
        print(f"{ts:>5} {nf:>7} {cin:>4}x{cout:<5} {rb.P:>9} | {t_f*1e3:8.3f} {fl/t_f/1e12:6.1f} | {t_d*1e3:8.3f} {fl/t_d/1e12:6.1f} | {t_w*1e3:8.3f} {fl/t_w/1e12:6.1f}")


if __name__ == "__main__":
    main()
