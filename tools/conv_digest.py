#!/usr/bin/env python
"""sha256 of the tile-compacted kernel's output on fixed inputs (bench scene's stride-1 / stride-2 maps): two builds of the
library must print the same digests when a change claims bit-identical results.
    USC3D_LIB=build/ablate/<name>.so python tools/conv_digest.py"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unscene3d_amd import MinkowskiEngine as ME  # noqa: E402
from unscene3d_amd import ops  # noqa: E402
from unscene3d_amd.synthetic import make_scene  # noqa: E402

dev = torch.device("cuda:0")
sc = make_scene(2000, target_voxels=150000)
c3, _, _ = ME.utils.sparse_quantize(sc["xyz"], quantization_size=0.02, return_index=True, return_inverse=True, device="cuda:0")
coords = torch.cat([torch.zeros((c3.shape[0], 1), dtype=torch.int32, device=dev), c3], 1).contiguous()
x = ME.SparseTensor(features=torch.zeros(coords.shape[0], 3, device=dev), coordinates=coords, device=dev)
cm = x.coordinate_manager
cm.stride_map(1)
for ts, (cin, cout) in ((1, (96, 96)), (1, (128, 96)), (1, (96, 128)), (2, (96, 96))):
    n = cm.coord_map(ts).n
    nbr = cm.cube_map(ts)["nbr"]
    g = torch.Generator(device="cpu").manual_seed(ts * 1000 + cin + cout)
    xin = torch.randn(n, cin, generator=g).to(dev)
    W = (torch.randn(27, cin, cout, generator=g) * 0.05).to(dev)
    out = ops.gather_gemm(xin, W, nbr, n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gather_gemm(xin, W, nbr, n)
    e1.record()
    torch.cuda.synchronize()
    print(f"stride {ts} rows {n} {cin}->{cout}: sha256 {hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]}  {e0.elapsed_time(e1) * 100:.1f} us per launch")
