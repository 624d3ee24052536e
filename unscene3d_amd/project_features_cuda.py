"""2D -> 3D feature projection.

Two layers, both on the HIP library (no CPU path):

* the reference extension's operator surface, argument for argument —
  `project_features_cuda(...)` / `unproject_depth_images(...)`
  (utils/cuda_utils/project_image_cuda.cpp:10-31): in-place accumulation into caller-allocated tensors, dense
  int64 occupancy grid as input;
* `Project2DFeaturesCUDA` (utils/cuda_utils/raycast_image.py:18-77): same constructor / forward and return
  values, but the occupancy comes from the coordinate hash (no dense grid is built), the hit pixels are reduced
  per voxel in a fixed order, and `fuse_frame` folds the caller's running mean over frames
  (pseudo_masks/unscene3d_pseudo_main.py:303-330) into the same kernel.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from ._lib import check, lib
from .ops import _chk, _ptr, _stream


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    return t.to(torch.float32).contiguous()


def raycast_first_hit_dense(occupancy_3d, views, intrinsics, height, width, depth_min, depth_max, ray_increment,
                            n_rows, want_seg=True):
    """occupancy i64[B,dz,dy,dx] -> (hit i32[B,V,H,W], seg i64[B*V*H*W] or None)."""
    ops.require_device()
    _chk(occupancy_3d, torch.int64, "occupancy_3D")
    B, dz, dy, dx = occupancy_3d.shape
    V = views.shape[1]
    views, intrinsics = _f32c(views, "viewMatrixInv"), _f32c(intrinsics, "intrinsicParams")
    dev = occupancy_3d.device
    hit = torch.empty((B, V, height, width), dtype=torch.int32, device=dev)
    seg = torch.empty(B * V * height * width, dtype=torch.int64, device=dev) if want_seg else None
    check(lib.usc_raycast_first_hit_dense(_ptr(occupancy_3d), dz, dy, dx, n_rows, _ptr(views), _ptr(intrinsics), B, V,
                                          height, width, depth_min, depth_max, ray_increment, _ptr(hit),
                                          _ptr(seg) if want_seg else None, _stream()), "usc_raycast_first_hit_dense")
    return hit, seg


def brick_mask(cmap: ops.CoordMap, shift):
    """Free-space filter for the ray cast: (mask u32[B, words], (bricks_x, bricks_y, bricks_z)) — one bit per
    8^3-voxel brick of the shifted grid that holds a voxel.  Build once per scene."""
    B = shift.shape[0]
    ext = (cmap.coords[:, 1:].amax(0) - shift.amin(0) + 1).tolist()
    bricks = tuple((int(e) + 7) // 8 for e in ext)
    words = (bricks[0] * bricks[1] * bricks[2] + 31) // 32
    mask = torch.empty((B, words), dtype=torch.int32, device=cmap.coords.device)
    check(lib.usc_brick_mask_build(_ptr(cmap.coords), cmap.n, _ptr(shift), B, *bricks, _ptr(mask), _stream()),
          "usc_brick_mask_build")
    return mask, bricks


def raycast_first_hit_map(cmap: ops.CoordMap, shift, views, intrinsics, height, width, depth_min, depth_max,
                          ray_increment, want_seg=True, bricks=None):
    """cmap: stride-1 coordinate map; shift i32[B,3] per-batch minimum coordinate; views already shifted;
    bricks: optional result of `brick_mask` (same hits, fewer hash probes)."""
    ops.require_device()
    _chk(shift, torch.int32, "shift")
    B, V = views.shape[0], views.shape[1]
    if shift.shape != (B, 3):
        raise RuntimeError("shift must be int32 [batch,3]")
    views, intrinsics = _f32c(views, "view_matrix"), _f32c(intrinsics, "intrinsic_params")
    dev = cmap.coords.device
    hit = torch.empty((B, V, height, width), dtype=torch.int32, device=dev)
    seg = torch.empty(B * V * height * width, dtype=torch.int64, device=dev) if want_seg else None
    bm, bd = bricks if bricks is not None else (None, (0, 0, 0))
    check(lib.usc_raycast_first_hit_map(_ptr(cmap.table_keys), _ptr(cmap.table_vals), cmap.cap, cmap.n, _ptr(shift),
                                        _ptr(bm) if bm is not None else None, bd[0], bd[1], bd[2],
                                        _ptr(views), _ptr(intrinsics), B, V, height, width, depth_min, depth_max,
                                        ray_increment, _ptr(hit), _ptr(seg) if want_seg else None, _stream()),
          "usc_raycast_first_hit_map")
    return hit, seg


_MODES = {"mean": 0, "fuse": 1, "accumulate": 2}


def project_reduce(feats, seg, n_rows, out, num=None, mode="mean"):
    """feats f32[n_pix,C]; seg i64[n_pix] in [0,n_rows] (n_rows = miss) or its SegmentCSR over n_rows+1 segments;
    out f32[n_rows,C]; num i32[n_rows]."""
    _chk(feats, torch.float32, "encoded_2d_features")
    _chk(out, torch.float32, "projected_features")
    if num is not None:
        _chk(num, torch.int32, "mapping2dto3d_num")
    csr = seg if isinstance(seg, ops.SegmentCSR) else ops.segment_csr(seg, n_rows + 1)
    if feats.shape[0] != csr.order.shape[0] or csr.S != n_rows + 1 or out.shape != (n_rows, feats.shape[1]):
        raise RuntimeError("project_reduce: shape mismatch")
    check(lib.usc_project_reduce(_ptr(feats), feats.shape[1], _ptr(csr.order), _ptr(csr.seg_off), n_rows, _MODES[mode],
                                 _ptr(out), _ptr(num) if num is not None else None, _stream()), "usc_project_reduce")
    return out, num


# --------------------------------------------------------------------------------------------------------------
# the extension's operator surface
def project_features_cuda(encoded_2d_features, occupancy_3D, viewMatrixInv, intrinsicParams, opts, mapping2dto3d_num,
                          projected_features, pred_mode_t):
    """In place, returns None — like the reference operator (project_image_cuda_kernel.cu:190-246).
    opts (host tensor): width, height, depth_min, depth_max, ray_increment; pred_mode_t: host bool tensor."""
    B, V, H, W, C = encoded_2d_features.shape
    o = [float(v) for v in opts.tolist()]
    width, height = int(o[0] + 0.5), int(o[1] + 0.5)
    if (height, width) != (H, W):
        raise RuntimeError(f"project_features_cuda: opts say {height}x{width}, features are {H}x{W}")
    n_rows = projected_features.shape[0]
    pred_mode = bool(pred_mode_t.reshape(-1)[0].item())
    hit, seg = raycast_first_hit_dense(occupancy_3D, viewMatrixInv, intrinsicParams, H, W, o[2], o[3], o[4], n_rows,
                                       want_seg=not pred_mode)
    if pred_mode:
        _chk(encoded_2d_features, torch.int32, "encoded_2d_features (prediction mode)")
        _chk(projected_features, torch.int32, "projected_features (prediction mode)")
        check(lib.usc_project_predictions(_ptr(encoded_2d_features), C, _ptr(hit), B * V * H * W,
                                          _ptr(projected_features), _stream()), "usc_project_predictions")
    else:
        _chk(encoded_2d_features, torch.float32, "encoded_2d_features")
        project_reduce(encoded_2d_features.view(-1, C), seg, n_rows, projected_features, mapping2dto3d_num,
                       mode="accumulate")


def unproject_depth_images(depth_images, viewMatrixInv, intrinsicParams, batched_point_cloud):
    """depth f32[V,H,W] -> batched_point_cloud f32[V*H*W,5] filled in place for depth > 0
    (project_image_cuda_kernel.cu:249-323)."""
    ops.require_device()
    _chk(depth_images, torch.float32, "depth_images")
    _chk(batched_point_cloud, torch.float32, "batched_point_cloud")
    V, H, W = depth_images.shape
    views, intr = _f32c(viewMatrixInv, "viewMatrixInv"), _f32c(intrinsicParams, "intrinsicParams")
    if batched_point_cloud.numel() != V * H * W * 5:
        raise RuntimeError("batched_point_cloud must have V*H*W rows of 5")
    check(lib.usc_unproject_depth(_ptr(depth_images), _ptr(views), _ptr(intr), V, H, W, _ptr(batched_point_cloud),
                                  _stream()), "usc_unproject_depth")


# --------------------------------------------------------------------------------------------------------------
class Project2DFeaturesCUDA(nn.Module):
    """Same interface as the reference module (utils/cuda_utils/raycast_image.py:18-77).  `config` is only read
    for `config.data.ignore_label` in prediction mode."""

    def __init__(self, width, height, voxel_size, config=None, depth_min=0.1, depth_max=4.0):
        super().__init__()
        self.image_width = width
        self.image_height = height
        self.voxel_size = voxel_size
        self.ray_increment = voxel_size / 2.
        self.config = config
        self.depth_min = depth_min / voxel_size
        self.depth_max = depth_max / voxel_size
        self._scene = None

    def _scene_state(self, coords):
        """Coordinate hash + per-batch shift of one scene, cached while the same coordinate tensor comes back
        (the caller projects 100-300 frames onto one scene)."""
        key = (coords.data_ptr(), tuple(coords.shape), coords._version)
        if self._scene is not None and self._scene[0] == key:
            return self._scene[1:4]
        c32 = coords.to(torch.int32).contiguous()
        batch_size = int(c32[-1, 0].item()) + 1
        shift = torch.stack([c32[c32[:, 0] == b, 1:].amin(0) for b in range(batch_size)]).contiguous()
        cmap, _, _ = ops.coordmap_build(c32)
        if cmap.n != c32.shape[0]:
            raise RuntimeError("Project2DFeaturesCUDA: duplicate voxel coordinates")
        bricks = brick_mask(cmap, shift)
        self._scene = (key, cmap, shift, bricks, coords)   # holding `coords` keeps its address from being reused
        return cmap, shift, bricks

    def _cast(self, encoded_2d_features, coords, view_matrix, intrinsic_params, want_seg):
        if not coords.is_cuda:
            raise RuntimeError("Project2DFeaturesCUDA needs CUDA tensors (there is no CPU path)")
        B, V, H, W, _ = encoded_2d_features.shape
        if (H, W) != (self.image_height, self.image_width):
            raise RuntimeError(f"features are {H}x{W}, the projecter was built for "
                               f"{self.image_height}x{self.image_width}")
        cmap, shift, bricks = self._scene_state(coords)
        local_views = view_matrix.detach().to(torch.float32).clone()
        local_views[:, :, :3, 3] -= shift.to(torch.float32)[:, None, :]
        hit, seg = raycast_first_hit_map(cmap, shift, local_views, intrinsic_params, H, W, self.depth_min,
                                         self.depth_max, self.ray_increment, want_seg=want_seg, bricks=bricks)
        return cmap.n, hit, seg

    def forward(self, encoded_2d_features, coords, view_matrix, intrinsic_params, pred_mode=False):
        C = encoded_2d_features.shape[-1]
        n, hit, seg = self._cast(encoded_2d_features, coords, view_matrix, intrinsic_params, want_seg=not pred_mode)
        dev = coords.device
        mapping2dto3d_num = torch.zeros(n, dtype=torch.int32, device=dev)
        if not pred_mode:
            feats = encoded_2d_features.to(torch.float32).contiguous().view(-1, C)
            projected = torch.empty((n, C), dtype=torch.float32, device=dev)
            project_reduce(feats, seg, n, projected, mapping2dto3d_num, mode="mean")
            return projected, mapping2dto3d_num
        ignore = self.config.data.ignore_label if self.config is not None else 255
        preds = encoded_2d_features.to(torch.int32).contiguous()
        projected = torch.full((n, C), ignore, dtype=torch.int32, device=dev)
        check(lib.usc_project_predictions(_ptr(preds), C, _ptr(hit), hit.numel(), _ptr(projected), _stream()),
              "usc_project_predictions")
        return projected.flatten().long(), mapping2dto3d_num

    def fuse_frame(self, scene_feats, encoded_2d_features, coords, view_matrix, intrinsic_params, hit_seg=None):
        """scene_feats[r] <- (scene_feats[r] + projected[r]) / 2 on the voxels this frame hits, in place — the
        caller's per-frame update (unscene3d_pseudo_main.py:311-313) without materialising `projected`.
        -> (mapping2dto3d_num, hit_seg); pass hit_seg back in to project a second feature map of the same frame
        (key and query features) without casting the rays again."""
        C = encoded_2d_features.shape[-1]
        _chk(scene_feats, torch.float32, "scene_feats")
        if hit_seg is None:
            n, _, seg = self._cast(encoded_2d_features, coords, view_matrix, intrinsic_params, want_seg=True)
            csr = ops.segment_csr(seg, n + 1)
        else:
            n, csr = hit_seg
        feats = encoded_2d_features.to(torch.float32).contiguous().view(-1, C)
        num = torch.empty(n, dtype=torch.int32, device=scene_feats.device)
        project_reduce(feats, csr, n, scene_feats, num, mode="fuse")
        return num, (n, csr)
