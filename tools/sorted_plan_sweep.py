#!/usr/bin/env python
"""Tile width x offset split sweep of the mask-sorted conv kernel on the bench scene's levels (forward launches):
the measurements the cost model in csrc/spconv_sorted.hip::plan_sorted is calibrated on.
Usage (GPU box): USC3D_SORTED_TUNE=1 python tools/sorted_plan_sweep.py > profiles/r02_sorted_plan.txt"""
import os
import sys

os.environ["USC3D_SORTED_TUNE"] = "1"
os.environ.setdefault("USC3D_CONV", "sorted-all")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unscene3d_amd import MinkowskiEngine as ME  # noqa: E402
from unscene3d_amd import ops  # noqa: E402
from unscene3d_amd.synthetic import make_scene  # noqa: E402


def timeit(fn, reps=20):
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
    ops.CONV_PATH = "sorted-all"
    sc = make_scene(2000, target_voxels=150000)
    c3, _, _ = ME.utils.sparse_quantize(sc["xyz"], quantization_size=0.02, return_index=True, return_inverse=True,
                                        device="cuda:0")
    coords = torch.cat([torch.zeros((c3.shape[0], 1), dtype=torch.int32, device=dev), c3], 1).contiguous()
    coords = ops.gather_rows_i32(coords, ops.spatial_order(coords))
    x = ME.SparseTensor(features=torch.zeros(coords.shape[0], 3, device=dev), coordinates=coords, device=dev)
    cm = x.coordinate_manager
    for ts in (1, 2, 4, 8):
        cm.stride_map(ts)
    shapes = {2: [(32, 32), (96, 96)], 4: [(64, 64), (128, 128), (192, 128)], 8: [(128, 128), (256, 256), (384, 256)],
              16: [(128, 256), (256, 256)]}
    print(f"{'level':>5} {'rows':>7} {'shape':>9} | us per (NB, G): the plan's own choice first")
    for ts, lst in shapes.items():
        n = cm.coord_map(ts).n
        nbr = cm.cube_map(ts)["nbr"]
        for cin, cout in lst:
            xin = torch.randn(n, cin, device=dev)
            W = torch.randn(27, cin, cout, device=dev) * 0.05
            os.environ.pop("USC3D_SORTED_NB", None)
            os.environ.pop("USC3D_SORTED_G", None)
            own = timeit(lambda: ops.gather_gemm(xin, W, nbr, n))
            row = [f"plan {own:6.1f}"]
            for nb in (4, 3, 2, 1):
                if (cout // 32) % nb:
                    continue
                for g in (1, 2, 3, 4, 5, 7, 9, 14, 27):
                    os.environ["USC3D_SORTED_NB"] = str(nb)
                    os.environ["USC3D_SORTED_G"] = str(g)
                    row.append(f"({nb},{g:2d}) {timeit(lambda: ops.gather_gemm(xin, W, nbr, n)):6.1f}")
            print(f"{ts:5d} {n:7d} {cin:4d}x{cout:<4d} | " + "  ".join(row), flush=True)


if __name__ == "__main__":
    main()
