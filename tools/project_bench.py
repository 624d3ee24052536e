"""Per-frame cost of the 2D -> 3D feature projection at the reference's size (192 x 256 rays, 384-d features,
~150 k voxels, depth 0.1-4 m, ray increment 0.01 voxel) with the oracle timed beside it on a sub-sampled frame.

    python tools/project_bench.py [--frames 20] [--cpu-rays 96]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from unscene3d_amd import ops  # noqa: E402
from unscene3d_amd import project_features_cuda as P  # noqa: E402
from unscene3d_amd.synthetic import camera_views, room_voxels  # noqa: E402


def _time(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--cpu-rays", type=int, default=96, help="rays of the oracle's sample (0: skip)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dims = (150, 170, 130)
    coords = room_voxels(5, dims=dims, n_boxes=30)
    n = coords.shape[0]
    H, W, C = 192, 256, 384
    dmin, dmax, inc = 0.1 / 0.02, 4.0 / 0.02, 0.01
    views = camera_views(6, coords, a.frames, dims=dims)
    intr = np.array([[W * 0.9, W * 0.9, (W - 1) / 2, (H - 1) / 2]], np.float32)
    c_d = torch.from_numpy(coords).to(dev)
    proj = P.Project2DFeaturesCUDA(W, H, 0.02)
    v_d = torch.from_numpy(views).to(dev)
    k_d = torch.from_numpy(intr).to(dev)
    feats = torch.randn((1, 1, H, W, C), device=dev)
    scene = torch.zeros((n, C), device=dev)

    cmap, shift, bricks = proj._scene_state(c_d)
    lv = v_d.clone()
    lv[:, :, :3, 3] -= shift.float()[:, None, :]
    out = {"voxels": n, "rays": H * W, "channels": C}
    fr = [0]

    def cast():
        i = fr[0] % a.frames
        fr[0] += 1
        return P.raycast_first_hit_map(cmap, shift, lv[:, i:i + 1], k_d, H, W, proj.depth_min, proj.depth_max,
                                       proj.ray_increment, bricks=use_bricks[0])

    use_bricks = [None]
    out["raycast_no_brick_mask_ms"] = _time(cast, a.frames)
    use_bricks[0] = bricks
    out["raycast_ms"] = _time(cast, a.frames)
    hit, seg = cast()
    out["hit_fraction"] = float((hit >= 0).float().mean())
    out["csr_ms"] = _time(lambda: ops.segment_csr(seg, n + 1), 20)
    csr = ops.segment_csr(seg, n + 1)
    num = torch.empty(n, dtype=torch.int32, device=dev)
    out["reduce_fuse_ms"] = _time(lambda: P.project_reduce(feats.view(-1, C), csr, n, scene, num, mode="fuse"), 20)
    dense = torch.empty((n, C), device=dev)
    out["reduce_mean_full_ms"] = _time(lambda: P.project_reduce(feats.view(-1, C), csr, n, dense, num, mode="mean"), 20)

    def frame():
        i = fr[0] % a.frames
        fr[0] += 1
        proj.fuse_frame(scene, feats, c_d, v_d[:, i:i + 1], k_d)

    out["frame_ms"] = _time(frame, a.frames)
    out["frames_per_s"] = 1000.0 / out["frame_ms"]
    # algorithmic bytes of the fused reduction: hit pixels' features read once + hit rows of the scene read+written
    hit_px = int((hit >= 0).sum())
    rows = int((num > 0).sum())
    out["reduce_fuse_GBps"] = (hit_px * C * 4 + 2 * rows * C * 4) / out["reduce_fuse_ms"] / 1e6

    if a.cpu_rays:
        from oracle import project_ref as PR

        occ, shifts = PR.dense_occupancy(coords)
        sv = PR.shift_views(views[:, :1], shifts)
        w = a.cpu_rays
        k = intr.copy()
        t0 = time.perf_counter()
        PR.first_hit(occ, sv, k, w, 1, dmin, dmax, inc)        # one image row of `w` rays
        dt = time.perf_counter() - t0
        out["cpu_oracle_rays"] = w
        out["cpu_oracle_s"] = dt
        out["cpu_oracle_frame_s_extrapolated"] = dt * H * W / w
    print(json.dumps(out))


if __name__ == "__main__":
    main()
