"""ME.utils subset: sparse_quantize / sparse_collate on the HIP device.

Reference call sites: datasets/utils.py:403-408 (`sparse_quantize(coordinates=…,
features=…, ignore_label=…, return_index=True, return_inverse=True)`), :430
(`sparse_collate(coords, feats, labels)`), pseudo_masks/datasets/voxelizer.py:142.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops


def _to_dev(x, dtype, device):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(device=device, dtype=dtype).contiguous()


def sparse_quantize(coordinates, features=None, labels=None, ignore_label=-100, return_index=False,
                    return_inverse=False, return_maps_only=False, quantization_size=None, device="cuda"):
    """Distinct coordinates in first-occurrence order (+ unique_map / inverse_map).

    `coordinates` [N,D] already floored (np.floor(x / voxel), reference datasets/utils.py:403)
    unless `quantization_size` is given.  Runs on the HIP device."""
    dev = torch.device(device)
    c = coordinates
    if quantization_size is not None:
        cf = _to_dev(c, torch.float64, dev)
        ci = ops.voxel_floor(cf, float(quantization_size))
    else:
        ci = _to_dev(c, torch.float64, dev).floor().to(torch.int32) if not _is_int(c) else _to_dev(c, torch.int32, dev)
    if ci.shape[1] == 3:
        ci4 = torch.cat([torch.zeros((ci.shape[0], 1), dtype=torch.int32, device=dev), ci], dim=1).contiguous()
    else:
        ci4 = ci
    cmap, unique_idx, inverse = ops.coordmap_build(ci4, quant=1)
    if return_maps_only:
        return (unique_idx, inverse) if return_inverse else unique_idx
    out_c = ci[unique_idx]
    outs = [out_c]
    if features is not None:
        f = features if isinstance(features, torch.Tensor) else torch.from_numpy(np.asarray(features))
        outs.append(f.to(dev)[unique_idx])
    if labels is not None:
        l = labels if isinstance(labels, torch.Tensor) else torch.from_numpy(np.asarray(labels))
        outs.append(l.to(dev)[unique_idx])
    if return_index:
        outs.append(unique_idx)
    if return_inverse:
        outs.append(inverse)
    return outs[0] if len(outs) == 1 else tuple(outs)


def _is_int(c):
    if isinstance(c, np.ndarray):
        return np.issubdtype(c.dtype, np.integer)
    return not c.dtype.is_floating_point


def sparse_collate(coords, feats, labels=None, dtype=torch.int32, device=None):
    """Prepend the batch index and concatenate (reference datasets/utils.py:430)."""
    cs, fs, ls = [], [], []
    for b, c in enumerate(coords):
        c = c if isinstance(c, torch.Tensor) else torch.from_numpy(np.asarray(c))
        c = c.to(dtype)
        bcol = torch.full((c.shape[0], 1), b, dtype=dtype, device=c.device)
        cs.append(torch.cat([bcol, c], dim=1))
        f = feats[b]
        fs.append(f if isinstance(f, torch.Tensor) else torch.from_numpy(np.asarray(f)))
        if labels is not None:
            l = labels[b]
            ls.append(l if isinstance(l, torch.Tensor) else torch.from_numpy(np.asarray(l)))
    C, F = torch.cat(cs, 0), torch.cat(fs, 0)
    if device is not None:
        C, F = C.to(device), F.to(device)
    if labels is not None:
        L = torch.cat(ls, 0)
        return C, F, (L.to(device) if device is not None else L)
    return C, F
