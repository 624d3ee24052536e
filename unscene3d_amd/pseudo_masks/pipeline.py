"""Feeders of the NCut core on the pseudo-mask path (reference pseudo_masks/unscene3d_pseudo_main.py):
the 3D branch of `encode_scene_feats` (:332-348: CSC features of a coarser level carried to the input
voxels by exact 1-NN) and the save-time lift of segment masks to the full-resolution cloud (:649-667).
Both 1-NN searches run on the device (usc_knn1) instead of scipy's KD-tree.  The image branch of
`encode_scene_feats` (:287-330) projects per-frame 2D features onto the voxels and keeps a running mean."""
from __future__ import annotations

import numpy as np
import torch

from .. import ops


def encode_scene_feats_3d(model, sinput, resolution_scale=2):
    """Res16UNet34CMultiRes features of level `res_{resolution_scale}` for every input voxel."""
    with torch.no_grad():
        _, feature_maps = model(sinput)
        enc = feature_maps[f"res_{resolution_scale}"]
        lr_coords = enc.C[:, 1:].float().contiguous()
        hr_coords = sinput.C[:, 1:].float().contiguous()
        _, match = ops.knn1(hr_coords, lr_coords)
        return enc.F[match].detach()


def encode_scene_feats_2d(model, images, camera_poses, color_intrinsics, coords, projecter, attention=False):
    """Image branch of `encode_scene_feats` (unscene3d_pseudo_main.py:287-330).

    model(img[1,1,c,h,w]) -> (key_features, query_features), each [1,1,H,W,C] (the 2D backbone is the caller's);
    images [1,n_frames,c,h,w], camera_poses [1,n_frames,4,4], color_intrinsics [1,4]; coords int[n,4];
    projecter: `Project2DFeaturesCUDA`.  One ray cast per frame serves both feature maps; every frame's features
    are reduced per voxel and folded into the running mean in one kernel (`fuse_frame`).
    -> scene_key_feats (and scene_query_feats when `attention`), f32[n,C]."""
    n = coords.shape[0]
    scene_key = scene_query = None
    with torch.no_grad():
        for i, img in enumerate(images[0]):
            key_features, query_features = model(img.unsqueeze(0).unsqueeze(0))
            if scene_key is None:
                C = key_features.shape[-1]
                scene_key = torch.zeros((n, C), dtype=torch.float32, device=coords.device)
                scene_query = torch.zeros_like(scene_key) if attention else None
            view = camera_poses[0, i].unsqueeze(0).unsqueeze(0)
            _, cast = projecter.fuse_frame(scene_key, key_features, coords, view, color_intrinsics)
            if attention:
                projecter.fuse_frame(scene_query, query_features, coords, view, color_intrinsics, hit_seg=cast)
    return (scene_key, scene_query) if attention else scene_key


def masks_to_full_resolution(voxel_coords: torch.Tensor, full_res_coords, voxel_size: float, segment_ids, bipartitions):
    """Per-voxel segment ids / masks -> full-resolution points by 1-NN against voxel centres (+0.5)."""
    dev = voxel_coords.device
    ref = (voxel_coords[:, -3:].float() + 0.5).contiguous()
    query = torch.as_tensor(np.asarray(full_res_coords) / voxel_size, dtype=torch.float32, device=dev).contiguous()
    _, match = ops.knn1(query, ref)
    match = match.cpu().numpy()
    return np.asarray(segment_ids)[match], np.asarray(bipartitions)[match]
