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


# ---------------------------------------------------------------------------------------------------------------------
# The other train-mode steps of the scene reader (reference datasets/freemask_semseg.py:334-406)
def affine_rows(table: torch.Tensor, M=None, t=None) -> torch.Tensor:
    """table[:, :3] <- table[:, :3] @ M^T + t in place (f64 arithmetic, one rounding; `usc_affine_rows`)."""
    ops.require_device()
    if table.dim() != 2 or table.shape[1] < 3 or not table.is_contiguous() or table.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("affine_rows: needs a contiguous f32/f64 [N, >=3] device tensor")
    M = np.ascontiguousarray(np.eye(3) if M is None else M, np.float64).reshape(9)
    t = np.ascontiguousarray(np.zeros(3) if t is None else t, np.float64).reshape(3)
    check(lib.usc_affine_rows(_ptr(table), int(table.dtype == torch.float64), table.shape[0], table.shape[1],
                              M.ctypes.data, t.ctypes.data, _stream()), "usc_affine_rows")
    return table


def center_and_shift(coords: torch.Tensor) -> torch.Tensor:
    """`coordinates -= coordinates.mean(0); coordinates += np.random.uniform(min, max) / 2` (:335-338), in place.
    The three uniform numbers come from numpy's global generator with the reference's own call; the sums / min / max
    are one read-back of nine scalars.  Two roundings like numpy (f32 subtract, then f64 add rounded to f32), and the
    column mean is numpy's: an f32 sum in row order (`usc_colsum_sequential`) divided by N in f32 — the centred
    coordinates are the reference's bit for bit (f32 tables; an f64 table takes the tree sum)."""
    if coords.dtype == torch.float32:
        sums = torch.empty(3, dtype=torch.float32, device=coords.device)
        check(lib.usc_colsum_sequential(_ptr(coords), coords.shape[0], coords.shape[1], 3, _ptr(sums), _stream()),
              "usc_colsum_sequential")
        first = sums
    else:
        first = coords[:, :3].mean(0).float()
    stats = torch.stack([first, coords[:, :3].amin(0).float(), coords[:, :3].amax(0).float()]).cpu().numpy()
    mean = (stats[0] / np.float32(coords.shape[0])).astype(np.float32) if coords.dtype == torch.float32 else stats[0]
    affine_rows(coords, t=-mean.astype(np.float64))
    lo, hi = (stats[1] - mean).astype(np.float32), (stats[2] - mean).astype(np.float32)   # min/max of the centred cloud
    shift = np.random.uniform(lo, hi) / 2
    return affine_rows(coords, t=shift)


def flip_axis(coords: torch.Tensor, axis: int) -> torch.Tensor:
    """`coordinates[:, i] = np.max(coordinates[:, i]) - coordinates[:, i]` (:348-351), in place."""
    cmax = float(coords[:, axis].amax().item())
    M, t = np.eye(3), np.zeros(3)
    M[axis, axis], t[axis] = -1.0, np.float64(np.float32(cmax))
    return affine_rows(coords, M, t)


def rotation_about_axis(axis, angle: float) -> np.ndarray:
    """Rodrigues rotation matrix (volumentations' RotateAroundAxis3d rotates points and normals by it)."""
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


class VolumeAugmentations:
    """The pipeline of conf/augmentation/volumentations_aug.yaml — Scale3d(+-0.1 per axis) and RotateAroundAxis3d about
    z (+-pi), y and x (+-pi/24), every transform `always_apply` — applied to device tables as ONE composed affine
    pass over the points and one over the normals.  volumentations is a third-party package that is not in the
    reference tree: the transforms follow its published definitions, the random draws (python's `random.uniform`, one
    per scale axis, one per rotation) are this module's own order — RNG stream UNPINNED against the package."""
    transforms = ("Scale3d", "RotateAroundAxis3d[z]", "RotateAroundAxis3d[y]", "RotateAroundAxis3d[x]")

    def __init__(self, scale_limit=0.1, rot_z=np.pi, rot_y=np.pi / 24, rot_x=np.pi / 24):
        self.scale_limit = scale_limit
        self.rot = (((0, 0, 1), rot_z), ((0, 1, 0), rot_y), ((1, 0, 0), rot_x))

    def draw(self):
        import random
        scale = np.array([1.0 + random.uniform(-self.scale_limit, self.scale_limit) for _ in range(3)])
        R = np.eye(3)
        for axis, limit in self.rot:
            R = rotation_about_axis(axis, random.uniform(-limit, limit)) @ R
        return scale, R

    def __call__(self, points, normals, features, labels, params=None):
        scale, R = self.draw() if params is None else params
        affine_rows(points, R @ np.diag(scale))          # points: scaled, then rotated
        if normals is not None:
            affine_rows(normals, R)                      # normals: rotated only
        return {"points": points, "normals": normals, "features": features, "labels": labels}


class ColorAugmentations:
    """conf/augmentation/albumentations_aug.yaml: RandomBrightnessContrast(+-0.2, +-0.2, brightness_by_max) and
    RGBShift(+-20 per channel) on the uint8 pseudo image.  albumentations applies both to uint8 data through 256-entry
    tables (clip to [0, 255], truncate); here the tables are composed on the host (768 numbers) and applied together
    with the colour normalisation by one `usc_color_lut` pass.  Third-party definitions, own draw order: UNPINNED."""

    def __init__(self, brightness=0.2, contrast=0.2, shift=20.0):
        self.brightness, self.contrast, self.shift = brightness, contrast, shift

    def draw(self):
        import random
        alpha = 1.0 + random.uniform(-self.contrast, self.contrast)
        beta = random.uniform(-self.brightness, self.brightness)
        return alpha, beta, [random.uniform(-self.shift, self.shift) for _ in range(3)]

    def tables(self, params=None):
        alpha, beta, shifts = self.draw() if params is None else params
        v = np.arange(256, dtype=np.float32)
        bc = np.clip(v * np.float32(alpha) + np.float32(beta * 255.0), 0, 255).astype(np.uint8)
        return np.stack([np.clip(bc.astype(np.float32) + np.float32(s), 0, 255).astype(np.uint8) for s in shifts])


def color_tables_to_features(color: torch.Tensor, tables_u8, mean255, inv_std255) -> torch.Tensor:
    """uint8-truncated colours -> `tables_u8` (u8[3,256] or None = identity) -> (v - mean*255) / (std*255) in f32
    (albumentations.Normalize as the reference configures it, :408-409), one pass."""
    v = np.arange(256, dtype=np.float32) if tables_u8 is None else None
    lut = np.stack([((v if tables_u8 is None else tables_u8[c].astype(np.float32)) - np.float32(mean255[c]))
                    * np.float32(inv_std255[c]) for c in range(3)]).astype(np.float32)
    out = torch.empty((color.shape[0], 3), dtype=torch.float32, device=color.device)
    lut_d = torch.from_numpy(lut).to(color.device)
    check(lib.usc_color_lut(_ptr(color), color.shape[0], color.shape[1], _ptr(lut_d), _ptr(out), 3, _stream()),
          "usc_color_lut")
    return out
