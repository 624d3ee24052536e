"""Feeders of the NCut core on the pseudo-mask path (reference pseudo_masks/unscene3d_pseudo_main.py):
the 3D branch of `encode_scene_feats` (:332-348: CSC features of a coarser level carried to the input
voxels by exact 1-NN) and the save-time lift of segment masks to the full-resolution cloud (:649-667).
Both 1-NN searches run on the device (usc_knn1) instead of scipy's KD-tree."""
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


def masks_to_full_resolution(voxel_coords: torch.Tensor, full_res_coords, voxel_size: float, segment_ids, bipartitions):
    """Per-voxel segment ids / masks -> full-resolution points by 1-NN against voxel centres (+0.5)."""
    dev = voxel_coords.device
    ref = (voxel_coords[:, -3:].float() + 0.5).contiguous()
    query = torch.as_tensor(np.asarray(full_res_coords) / voxel_size, dtype=torch.float32, device=dev).contiguous()
    _, match = ops.knn1(query, ref)
    match = match.cpu().numpy()
    return np.asarray(segment_ids)[match], np.asarray(bipartitions)[match]
