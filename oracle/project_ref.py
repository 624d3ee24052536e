"""TEST INFRASTRUCTURE ONLY — CPU (numpy, fp32) restatement of the reference's 2D->3D feature projection.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (unscene3d_amd/project_features_cuda.py -> usc_raycast_* / usc_project_*) never does.

PARITY UNPINNED: the reference implements this path only as a CUDA extension
(utils/cuda_utils/project_image_cuda_kernel.cu) with no test, fixture or golden vector of its own, and it cannot
run in a container without an NVIDIA device, so this restatement is anchored on the source text alone:

* ray set-up            project_image_cuda_kernel.cu:124-131 (kinectProjToCamera: include/cudaUtil.h:104-117,
                        float4x4 * float3 / float4: include/cuda_SimpleMatrixUtil.h:626-642, row-major 4x4)
* ray march             project_image_cuda_kernel.cu:24-56   (round half away from zero: make_int3(p + sign(p)*0.5),
                        include/cutil_math.h:31-33,179-182; occupancy value 0 == "empty", so voxel row 0 never hits)
* feature accumulation  project_image_cuda_kernel.cu:58-65   (atomicAdd per channel; order is unspecified in the
                        reference — restated here in ascending pixel order, which is what the HIP path guarantees)
* prediction mode       project_image_cuda_kernel.cu:68-109  (atomicMax over int predictions)
* wrapper               utils/cuda_utils/raycast_image.py:18-77 (per-batch min shift, dense occupancy grid,
                        divide by count + 10e-5)
* frame fusion          pseudo_masks/unscene3d_pseudo_main.py:311-330 (running pairwise mean on the hit voxels)
* depth unprojection    project_image_cuda_kernel.cu:249-290

Arithmetic: fp32, every multiply and add rounded separately (no fused multiply-add), 1/sqrt instead of the
reference's approximate rsqrtf (2 ulp, not reproducible off an NVIDIA device).
"""
from __future__ import annotations

import numpy as np

F = np.float32


def _normalize(x, y, z):
    inv = F(1.0) / np.sqrt(x * x + y * y + z * z)
    return x * inv, y * inv, z * inv


def ray_setup(view_inv, intr, width, height, depth_min, depth_max):
    """view_inv f32[4,4] (row major), intr f32[4] = fx, fy, mx, my  ->  cam f32[3], dir f32[H,W,3], t0, t1 f32[H,W]."""
    m = np.asarray(view_inv, F).reshape(4, 4)
    fx, fy, mx, my = (F(v) for v in intr)
    dmin, dmax = F(depth_min), F(depth_max)
    ux = np.arange(width, dtype=F)[None, :].repeat(height, 0)
    uy = np.arange(height, dtype=F)[:, None].repeat(width, 1)
    depth = F(1.0) * (dmax - dmin) + dmin
    cx, cy, cz = _normalize(depth * ((ux - mx) / fx), depth * ((uy - my) / fy), np.full_like(ux, depth))
    cam = np.array([m[0, 3], m[1, 3], m[2, 3]], F)
    z0 = F(0.0)
    wx = m[0, 0] * cx + m[0, 1] * cy + m[0, 2] * cz + m[0, 3] * z0
    wy = m[1, 0] * cx + m[1, 1] * cy + m[1, 2] * cz + m[1, 3] * z0
    wz = m[2, 0] * cx + m[2, 1] * cy + m[2, 2] * cz + m[2, 3] * z0
    dx, dy, dz = _normalize(wx, wy, wz)
    to_len = F(1.0) / cz
    return cam, np.stack([dx, dy, dz], -1), to_len * dmin, to_len * dmax


def _round_away(p):
    s = (p > 0).astype(F) - (p < 0).astype(F)
    return np.trunc(p + s * F(0.5)).astype(np.int64)


def first_hit(occ, views, intr, width, height, depth_min, depth_max, ray_inc):
    """occ i64[B,dz,dy,dx] (0 = empty), views f32[B,V,4,4], intr f32[B,4] -> hit i32[B,V,H,W], -1 = no hit."""
    B, dz, dy, dx = occ.shape
    V = views.shape[1]
    hit = np.full((B, V, height, width), -1, np.int32)
    inc = F(ray_inc)
    for b in range(B):
        grid = occ[b]
        for v in range(V):
            cam, d, t, t1 = ray_setup(views[b, v], intr[b], width, height, depth_min, depth_max)
            t = t.reshape(-1).copy()
            t1 = t1.reshape(-1)
            d = d.reshape(-1, 3)
            res = np.full(t.shape[0], -1, np.int32)
            live = np.nonzero(t < t1)[0]
            while live.size:
                tl = t[live]
                px = _round_away(cam[0] + tl * d[live, 0])
                py = _round_away(cam[1] + tl * d[live, 1])
                pz = _round_away(cam[2] + tl * d[live, 2])
                inb = (px >= 0) & (py >= 0) & (pz >= 0) & (px < dx) & (py < dy) & (pz < dz)
                idx = np.zeros(live.size, np.int64)
                idx[inb] = grid[pz[inb], py[inb], px[inb]]
                idx = idx.astype(np.int32)
                got = idx != 0
                res[live[got]] = idx[got]
                t[live] = tl + inc
                live = live[~got]
                live = live[t[live] < t1[live]]
            hit[b, v] = res.reshape(height, width)
    return hit


def dense_occupancy(coords):
    """coords i[n,4] (b,x,y,z) -> (occ i64[B,dz,dy,dx] holding the voxel row, per-batch shift i[B,3]);
    raycast_image.py:34-55."""
    coords = np.asarray(coords)
    B = int(coords[-1, 0]) + 1
    local = coords.copy()
    shifts = np.zeros((B, 3), coords.dtype)
    for b in range(B):
        m = coords[:, 0] == b
        shifts[b] = coords[m, 1:].min(0)
        local[m, 1:] -= shifts[b]
    dims = local[:, 1:].max(0) + 1
    occ = np.zeros((B, dims[2], dims[1], dims[0]), np.int64)
    occ[local[:, 0], local[:, 3], local[:, 2], local[:, 1]] = np.arange(coords.shape[0])
    return occ, shifts


def shift_views(views, shifts):
    out = np.array(views, F, copy=True)
    for b in range(out.shape[0]):
        out[b, :, :3, 3] -= shifts[b].astype(F)
    return out


def project_features(feats, hit, n_voxels):
    """feats f32[B,V,H,W,C], hit i32[B,V,H,W] -> (projected f32[n,C] = sum / (count + 1e-4), count i32[n])."""
    C = feats.shape[-1]
    flat = hit.reshape(-1)
    pix = np.nonzero(flat >= 0)[0]
    num = np.zeros(n_voxels, np.int32)
    np.add.at(num, flat[pix], 1)
    acc = np.zeros((n_voxels, C), F)
    np.add.at(acc, flat[pix], feats.reshape(-1, C)[pix])          # sequential fp32 adds in ascending pixel order
    return acc / (num.astype(F)[:, None] + F(10e-5)), num


def project_predictions(preds, hit, n_voxels, ignore_label):
    C = preds.shape[-1]
    flat = hit.reshape(-1)
    pix = np.nonzero(flat >= 0)[0]
    out = np.full((n_voxels, C), ignore_label, np.int32)
    np.maximum.at(out, flat[pix], preds.reshape(-1, C)[pix].astype(np.int32))
    return out


def fuse_frame(scene, projected, num):
    """unscene3d_pseudo_main.py:311-313: scene[hit] = mean(stack(scene[hit], projected[hit]))."""
    m = num > 0
    scene = scene.copy()
    scene[m] = (scene[m] + projected[m]) / F(2.0)
    return scene


def unproject_depth(depth, views, intr):
    """depth f32[V,H,W], views f32[V,4,4], intr f32[V,4] -> f32[V*H*W,5] (view, pixel index, x, y, z); rows of
    invalid pixels (depth <= 0) stay as the caller initialised them (zeros here)."""
    V, H, W = depth.shape
    out = np.zeros((V * H * W, 5), F)
    for v in range(V):
        m = np.asarray(views[v], F)
        fx, fy, mx, my = (F(a) for a in intr[v])
        x = np.arange(W, dtype=F)[None, :].repeat(H, 0)
        y = np.arange(H, dtype=F)[:, None].repeat(W, 1)
        d = depth[v].astype(F)
        px = (x - mx) * d / fx
        py = (y - my) * d / fy
        wx = m[0, 0] * px + m[0, 1] * py + m[0, 2] * d + m[0, 3]
        wy = m[1, 0] * px + m[1, 1] * py + m[1, 2] * d + m[1, 3]
        wz = m[2, 0] * px + m[2, 1] * py + m[2, 2] * d + m[2, 3]
        idx = (v * H * W + np.arange(H * W)).reshape(H, W)
        ok = d > 0
        rows = idx[ok]
        out[rows, 0] = F(v)
        out[rows, 1] = idx[ok].astype(F)
        out[rows, 2], out[rows, 3], out[rows, 4] = wx[ok], wy[ok], wz[ok]
    return out
