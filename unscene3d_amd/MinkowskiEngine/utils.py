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
    if dev.type == "cpu":
        return _sparse_quantize_host(coordinates, features, labels, return_index, return_inverse, return_maps_only,
                                     quantization_size)
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


def _sparse_quantize_host(coordinates, features, labels, return_index, return_inverse, return_maps_only,
                          quantization_size):
    """device="cpu": the HIP-FREE path (usc_voxel_floor_f64_host / usc_unique_coords_host) — what a forked DataLoader
    worker of the reference's collate (datasets/utils.py:403-414, conf/data/indoor.yaml:24) can call; it touches no
    device and returns CPU tensors, bit-equal to the device path's indices."""
    from .._lib import check, lib
    c = coordinates.detach().cpu().numpy() if isinstance(coordinates, torch.Tensor) else np.asarray(coordinates)
    n, d = c.shape
    if quantization_size is not None or not np.issubdtype(c.dtype, np.integer):
        cf = np.ascontiguousarray(c, dtype=np.float64)
        ci = np.empty((n, d), dtype=np.int32)
        if d != 3:
            raise RuntimeError("sparse_quantize(device='cpu'): floating-point coordinates must be [N, 3]")
        check(lib.usc_voxel_floor_f64_host(cf.ctypes.data, n, float(quantization_size or 1.0), ci.ctypes.data),
              "usc_voxel_floor_f64_host")
    else:
        ci = np.ascontiguousarray(c, dtype=np.int32)
    uniq = np.empty(n, dtype=np.int64)
    inv = np.empty(n, dtype=np.int64)
    import ctypes
    n_out = ctypes.c_int64(0)
    check(lib.usc_unique_coords_host(ci.ctypes.data, n, d, uniq.ctypes.data, inv.ctypes.data, ctypes.byref(n_out)),
          "usc_unique_coords_host")
    unique_idx = torch.from_numpy(uniq[:n_out.value].copy())
    inverse = torch.from_numpy(inv)
    if return_maps_only:
        return (unique_idx, inverse) if return_inverse else unique_idx
    outs = [torch.from_numpy(ci)[unique_idx]]
    for extra in (features, labels):
        if extra is not None:
            t = extra if isinstance(extra, torch.Tensor) else torch.from_numpy(np.asarray(extra))
            outs.append(t.cpu()[unique_idx])
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
