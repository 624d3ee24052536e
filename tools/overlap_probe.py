#!/usr/bin/env python
"""How much of a coarse level's backward could overlap: the input-gradient convolutions of a level form a dependent chain,
the weight gradients only need dy.  Times R convolutions' (dgrad, wgrad) pairs (a) on one stream, (b) dgrads on one
stream and wgrads on a second one with no joins in between (upper bound of what any co-scheduling can give).
Usage (GPU box): python tools/overlap_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unscene3d_amd import MinkowskiEngine as ME  # noqa: E402
from unscene3d_amd import ops  # noqa: E402
from unscene3d_amd.synthetic import make_scene  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    sc = make_scene(2000, target_voxels=150000)
    c3, _, _ = ME.utils.sparse_quantize(sc["xyz"], quantization_size=0.02, return_index=True, return_inverse=True, device="cuda:0")
    coords = torch.cat([torch.zeros((c3.shape[0], 1), dtype=torch.int32, device=dev), c3], 1).contiguous()
    coords = ops.gather_rows_i32(coords, ops.spatial_order(coords))
    x = ME.SparseTensor(features=torch.zeros(coords.shape[0], 3, device=dev), coordinates=coords, device=dev)
    cm = x.coordinate_manager
    for ts in (1, 2, 4, 8):
        cm.stride_map(ts)
    R = 8
    side = torch.cuda.Stream()
    for ts, cin, cout in ((16, 256, 256), (8, 256, 256), (8, 128, 128), (4, 128, 128), (4, 64, 64), (2, 32, 32)):
        n = cm.coord_map(ts).n
        nbr = cm.cube_map(ts)["nbr"]
        rb = cm.cube_rulebook(ts)
        xin = [torch.randn(n, cin, device=dev) for _ in range(R)]
        dy = [torch.randn(n, cout, device=dev) for _ in range(R)]
        Wt = [ops.weight_transpose(torch.randn(27, cin, cout, device=dev) * 0.05, True) for _ in range(R)]
        dW = [torch.zeros(27, cin, cout, device=dev) for _ in range(R)]

        def dgrads():
            for r in range(R):
                ops.gather_gemm(dy[r], Wt[r], nbr, n)

        def wgrads():
            for r in range(R):
                ops.wgrad(xin[r], dy[r], 27, rb.in_idx, rb.out_idx, rb.koff, into=dW[r])

        def serial():
            for r in range(R):
                ops.gather_gemm(dy[r], Wt[r], nbr, n)
                ops.wgrad(xin[r], dy[r], 27, rb.in_idx, rb.out_idx, rb.koff, into=dW[r])

        def two_streams():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                wgrads()
            dgrads()
            torch.cuda.current_stream().wait_stream(side)

        def t(fn, reps=10):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps / R * 1e3

        print(f"stride {ts:2d} n={n:6d} {cin}->{cout}: per conv  dgrad {t(dgrads):6.1f} us  wgrad {t(wgrads):6.1f} us  "
              f"one stream {t(serial):6.1f} us  two streams {t(two_streams):6.1f} us")


if __name__ == "__main__":
    main()
