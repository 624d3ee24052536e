#!/usr/bin/env python
"""Where a tile of the tile-compacted kernel spends its cycles: runs forward launches of one shape on the bench scene's
stride-1 map with a developer build that counts phases (tools/build_ablate.sh phase "-DUSC_PHASE_STATS") and prints the
breakdown.  Usage (GPU box):
  bash tools/build_ablate.sh phase "-DUSC_PHASE_STATS" && USC3D_LIB=build/ablate/phase.so python tools/compact_phase.py"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unscene3d_amd import MinkowskiEngine as ME  # noqa: E402
from unscene3d_amd import ops  # noqa: E402
from unscene3d_amd._lib import LIB_PATH  # noqa: E402
from unscene3d_amd.synthetic import make_scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=150000)
    ap.add_argument("--cin", type=int, default=96)
    ap.add_argument("--cout", type=int, default=96)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    raw = C.CDLL(LIB_PATH)
    raw.usc_phase_stats_read.argtypes = [C.c_void_p]
    dev = torch.device("cuda:0")
    sc = make_scene(2000, target_voxels=a.voxels)
    c3, _, _ = ME.utils.sparse_quantize(sc["xyz"], quantization_size=0.02, return_index=True, return_inverse=True, device="cuda:0")
    coords = torch.cat([torch.zeros((c3.shape[0], 1), dtype=torch.int32, device=dev), c3], 1).contiguous()
    x = ME.SparseTensor(features=torch.zeros(coords.shape[0], 3, device=dev), coordinates=coords, device=dev)
    cm = x.coordinate_manager
    ts = a.stride
    t = 1
    while t < ts:
        cm.stride_map(t)
        t *= 2
    n = cm.coord_map(ts).n
    nbr = cm.cube_map(ts)["nbr"]
    P = cm.cube_rulebook(ts).P
    xin = torch.randn(n, a.cin, device=dev)
    W = torch.randn(27, a.cin, a.cout, device=dev) * 0.05
    ops.gather_gemm(xin, W, nbr, n)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    raw.usc_phase_stats_read(buf)            # clear
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        ops.gather_gemm(xin, W, nbr, n)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    raw.usc_phase_stats_read(buf)
    v = [int(b) for b in buf]
    wgs, items = max(1, v[6]), max(1, v[7])
    waves = wgs * 8
    tile = (v[0] + v[1] + v[2]) / wgs
    fl = 2.0 * P * a.cin * a.cout
    print(f"{n} rows, {a.cin}->{a.cout}, {P} pairs: {ms * 1e3:.1f} us per launch = {fl / ms / 1e9:.1f} TFLOP/s; "
          f"{wgs // a.reps} tiles per launch, {items / wgs:.1f} items per tile (slot efficiency {P * a.reps / (32.0 * items):.3f})")
    print(f"per tile (shader-clock cycles, mean): prologue {v[0] / wgs:.0f}, main loop (until the last wave is done) {v[1] / wgs:.0f}, "
          f"write-back {v[2] / wgs:.0f}  -> tile {tile:.0f}")
    print(f"per wave: main loop {v[8] / waves:.0f}, of it waiting for the ticket {v[4] / waves:.0f} ({100.0 * v[4] / max(1, v[8]):.1f} %), "
          f"flushing {v[5] / waves:.0f} ({100.0 * v[5] / max(1, v[8]):.1f} %); idle at the tile's end barrier {v[3] / waves:.0f} "
          f"({100.0 * v[3] / max(1, v[8] + v[3]):.1f} % of main + idle)")
    print(f"per item: {v[8] / (items):.0f} wave-cycles... ({v[8] / waves / (items / waves):.0f} cycles per item per wave); MFMA issue floor per item = "
          f"{(a.cin // 2) * (a.cout // 32) * 64} pipe cycles")


if __name__ == "__main__":
    main()
