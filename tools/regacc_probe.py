#!/usr/bin/env python
"""EXPERIMENT: the register-accumulator conv kernel (csrc/spconv_regacc.hip) against the tile-compacted kernel on the bench
scene's large maps.  Usage (GPU box): python tools/regacc_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unscene3d_amd import MinkowskiEngine as ME  # noqa: E402
from unscene3d_amd import ops  # noqa: E402
from unscene3d_amd._lib import check, lib  # noqa: E402
from unscene3d_amd.synthetic import make_scene  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    sc = make_scene(2000, target_voxels=150000)
    c3, _, _ = ME.utils.sparse_quantize(sc["xyz"], quantization_size=0.02, return_index=True, return_inverse=True, device="cuda:0")
    coords = torch.cat([torch.zeros((c3.shape[0], 1), dtype=torch.int32, device=dev), c3], 1).contiguous()
    coords = ops.gather_rows_i32(coords, ops.spatial_order(coords, 5))
    x = ME.SparseTensor(features=torch.zeros(coords.shape[0], 3, device=dev), coordinates=coords, device=dev)
    cm = x.coordinate_manager
    cm.stride_map(1)
    for ts, cin, cout in ((1, 96, 96), (1, 128, 96), (1, 96, 128), (2, 96, 96), (2, 128, 96)):
        n = cm.coord_map(ts).n
        nbr = cm.cube_map(ts)["nbr"]
        perm, tmask = ops.rowsort(nbr)
        rb = cm.cube_rulebook(ts)
        xin = torch.randn(n, cin, device=dev)
        W = torch.randn(27, cin, cout, device=dev) * 0.05
        ws = torch.empty(lib.usc_spconv_regacc_ws_bytes(cin, cout, 27), dtype=torch.uint8, device=dev)
        out = torch.empty(n, cout, device=dev)

        def reg(wt=0, src=xin, Wt=W, o=out, ci=cin, co=cout):
            check(lib.usc_spconv_regacc_gemm(src.data_ptr(), n, ci, Wt.data_ptr(), 27, co, nbr.data_ptr(), perm.data_ptr(),
                                             tmask.data_ptr(), n, None, o.data_ptr(), 0, wt, ws.data_ptr(), ws.numel(), ops._stream()),
                  "usc_spconv_regacc_gemm")
            return o
        y0 = ops.gather_gemm(xin, W, nbr, n)
        y1 = reg().clone()
        err = float((y1 - y0).norm() / y0.norm())
        t0 = timeit(lambda: ops.gather_gemm(xin, W, nbr, n))
        t1 = timeit(lambda: reg())
        t0b = timeit(lambda: ops.gather_gemm(xin, W, nbr, n))
        fl = 2.0 * rb.P * cin * cout
        # input gradient: dy [n, cout] with the forward weights, transposed mode
        dy = torch.randn(n, cout, device=dev)
        dx0 = ops.gather_gemm(dy, W, nbr, n, w_transposed=True)
        dxo = torch.empty(n, cin, device=dev)
        dx1 = reg(1, dy, W, dxo, cout, cin).clone()
        errd = float((dx1 - dx0).norm() / dx0.norm())
        td0 = timeit(lambda: ops.gather_gemm(dy, W, nbr, n, w_transposed=True))
        td1 = timeit(lambda: reg(1, dy, W, dxo, cout, cin))
        print(f"stride {ts} n={n} {cin}->{cout}: fwd compact {t0:.1f}/{t0b:.1f} us ({fl/min(t0,t0b)/1e6:.1f} TF)  regacc {t1:.1f} us ({fl/t1/1e6:.1f} TF) rel {err:.1e} | "
              f"dgrad compact {td0:.1f} us  regacc {td1:.1f} us rel {errd:.1e}")


if __name__ == "__main__":
    main()
