"""Elastic distortion of the training augmentation on the device (reference
datasets/semseg.py:651-688, applied twice per training sample by freemask_semseg.py:356-361 with
(granularity, magnitude) = (0.2, 0.4) and (0.8, 1.6)).

The random part stays exactly the reference's: the noise grid is drawn from numpy's global generator with the same
call (`np.random.randn(*noise_dim, 3)`), so a seeded run consumes the same stream.  The grid is a few thousand cells:
its size and axes are computed with numpy on six scalars read back from the device, the box smoothing runs on the
device in f64 with one f32 rounding per pass like scipy.ndimage.convolve, and the per-point trilinear displacement
(`usc_elastic_displace`, f64 like scipy's RegularGridInterpolator) is one launch over the cloud."""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from .._lib import check, lib
from ..ops import _ptr, _stream


def noise_grid(coords_min, coords_max, granularity):
    """-> (noise_dim int[3], axes: three f64 arrays) exactly as the reference derives them from the f32 extent."""
    mn, mx = np.asarray(coords_min), np.asarray(coords_max)
    noise_dim = ((mx - mn) // granularity).astype(int) + 3
    axes = [np.linspace(d_min, d_max, d)
            for d_min, d_max, d in zip(mn - granularity, mn + granularity * (noise_dim - 2), noise_dim)]
    return noise_dim, axes


def smooth_noise(noise: torch.Tensor) -> torch.Tensor:
    """Two rounds of 3-tap box filters along x, y, z with zero padding; f64 sums of the f32(1/3)-weighted taps, rounded
    to f32 after every pass (scipy.ndimage.convolve on an f32 array)."""
    w = float(np.float32(1.0) / np.float32(3.0))
    x = noise
    for _ in range(2):
        for axis in range(3):
            d = x.double()
            pad = [0, 0] * (x.dim() - 1 - axis) + [1, 1]
            d = torch.nn.functional.pad(d, pad)
            n = x.shape[axis]
            a, b, c = d.narrow(axis, 0, n), d.narrow(axis, 1, n), d.narrow(axis, 2, n)
            x = (c * w + b * w + a * w).float()
    return x


def elastic_distortion(pointcloud: torch.Tensor, granularity: float, magnitude: float, noise=None) -> torch.Tensor:
    """pointcloud: device tensor [N, >= 3] (f32 or f64); the first three columns are displaced IN PLACE and the tensor
    is returned, like the reference.  noise: optional f32 [dx,dy,dz,3] (unsmoothed) instead of the numpy draw."""
    ops.require_device()
    if pointcloud.dim() != 2 or pointcloud.shape[1] < 3 or not pointcloud.is_contiguous():
        raise RuntimeError("pointcloud must be a contiguous [N, >=3] tensor")
    if pointcloud.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("pointcloud must be float32 or float64")
    xyz = pointcloud[:, :3]
    lo, hi = xyz.amin(0), xyz.amax(0)
    ends = torch.stack([lo, hi]).cpu().numpy()                     # the one read-back: six scalars
    noise_dim, axes = noise_grid(ends[0], ends[1], granularity)
    if noise is None:
        noise = np.random.randn(*noise_dim, 3).astype(np.float32)
    noise = torch.as_tensor(noise, dtype=torch.float32, device=pointcloud.device)
    if tuple(noise.shape) != (*noise_dim, 3):
        raise RuntimeError(f"noise must have shape {(*noise_dim, 3)}")
    noise = smooth_noise(noise).contiguous()
    ax = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(pointcloud.device) for a in axes]
    check(lib.usc_elastic_displace(_ptr(pointcloud), int(pointcloud.dtype == torch.float64), pointcloud.shape[0],
                                   pointcloud.shape[1], _ptr(noise), int(noise_dim[0]), int(noise_dim[1]),
                                   int(noise_dim[2]), _ptr(ax[0]), _ptr(ax[1]), _ptr(ax[2]), float(magnitude),
                                   _ptr(pointcloud), _stream()), "usc_elastic_displace")
    return pointcloud
