"""Synthetic ScanNet-shaped scenes — the workload generator for
bench.py, smoke() and the parity tests.  Pure numpy, deterministic in `seed`.

A scene is an axis-aligned room (aspect 8 : 6, height 2.8 m) with a floor, four
walls, 12-20 cuboids and 3-6 thin slabs.  Surface points are sampled uniformly by
area (≈3 points per 2 cm cell, N(0, 2 mm) jitter) and the room footprint is scaled
by bisection until 2 cm voxelisation yields `target_voxels` ± 2 % — i.e. surfaces
stay densely covered like a real ScanNet scan instead of being thinned out.
Outputs follow the dataset 9-tuple's columns that the hot path consumes
(reference datasets/freemask_semseg.py:434): xyz, colour features, raw xyz,
segment ids, object-wise pseudo masks, directed segment connectivity.
"""
from __future__ import annotations

import numpy as np

VOXEL = 0.02
_COLOR_MEAN = np.array([0.47793125906962, 0.4303257521323044, 0.3749598901421883]) * 255.0
_COLOR_STD = np.array([0.2834475483823543, 0.27566157565723015, 0.27018971370874995]) * 255.0


def _rects_of_box(lo, hi, obj, with_bottom=False):
    """Six (five) faces of an axis-aligned box as (origin, u, v, obj, face)."""
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    d = hi - lo
    ex, ey, ez = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
    faces = [
        (lo + ez, ex, ey),            # top
        (lo, ex, ez), (lo + ey, ex, ez),   # y- / y+
        (lo, ey, ez), (lo + ex, ey, ez),   # x- / x+
    ]
    if with_bottom:
        faces.append((lo, ex, ey))
    return [(o, u, v, obj, f) for f, (o, u, v) in enumerate(faces)]


def _layout(rng, s):
    """Rectangles of one room at footprint scale s."""
    W, D, H = 8.0 * s, 6.0 * s, 2.8
    rects = []
    obj = 0
    rects.append((np.zeros(3), np.array([W, 0, 0]), np.array([0, D, 0]), obj, 0))  # floor
    obj += 1
    walls = [
        (np.zeros(3), np.array([W, 0, 0]), np.array([0, 0, H])),
        (np.array([0, D, 0.0]), np.array([W, 0, 0]), np.array([0, 0, H])),
        (np.zeros(3), np.array([0, D, 0]), np.array([0, 0, H])),
        (np.array([W, 0, 0.0]), np.array([0, D, 0]), np.array([0, 0, H])),
    ]
    for o, u, v in walls:
        rects.append((o, u, v, obj, 0))
        obj += 1
    n_cub = int(rng.integers(12, 21))
    for _ in range(n_cub):
        e = rng.uniform(0.3, 2.0, 3) * min(1.0, s * 1.6)
        e[2] = min(e[2], H - 0.2)
        x0 = rng.uniform(0.05, max(W - e[0] - 0.05, 0.06))
        y0 = rng.uniform(0.05, max(D - e[1] - 0.05, 0.06))
        z0 = 0.0 if rng.random() < 0.75 else rng.uniform(0.5, max(H - e[2], 0.6))
        rects += _rects_of_box([x0, y0, z0], [x0 + e[0], y0 + e[1], min(z0 + e[2], H)], obj, with_bottom=z0 > 0)
        obj += 1
    n_slab = int(rng.integers(3, 7))
    for _ in range(n_slab):
        e = np.array([rng.uniform(0.6, 1.8), rng.uniform(0.4, 1.2), 0.04]) * min(1.0, s * 1.6)
        x0 = rng.uniform(0.05, max(W - e[0] - 0.05, 0.06))
        y0 = rng.uniform(0.05, max(D - e[1] - 0.05, 0.06))
        z0 = rng.uniform(0.4, 1.6)
        rects += _rects_of_box([x0, y0, z0], [x0 + e[0], y0 + e[1], z0 + e[2]], obj, with_bottom=True)
        obj += 1
    return rects, obj


def _sample(rects, rng, density):
    pts, objs, faces, tiles = [], [], [], []
    for o, u, v, obj, face in rects:
        lu, lv = np.linalg.norm(u), np.linalg.norm(v)
        n = int(np.ceil(lu * lv * density))
        if n == 0:
            continue
        a, b = rng.random(n), rng.random(n)
        p = o[None] + a[:, None] * u[None] + b[:, None] * v[None]
        p += rng.normal(0.0, 0.002, p.shape)
        pts.append(p)
        objs.append(np.full(n, obj, np.int64))
        faces.append(np.full(n, face, np.int64))
        tiles.append((np.floor(a * lu / 0.35).astype(np.int64) * 64 + np.floor(b * lv / 0.35).astype(np.int64)))
    return np.concatenate(pts), np.concatenate(objs), np.concatenate(faces), np.concatenate(tiles)


def _count_voxels(p):
    c = np.floor(p / VOXEL).astype(np.int64)
    key = (c[:, 0] + 4096) * (1 << 40) + (c[:, 1] + 4096) * (1 << 20) + (c[:, 2] + 4096)
    return np.unique(key).shape[0]


def make_scene(seed: int, target_voxels: int = 150_000, points_per_cell: float = 3.0, tol: float = 0.02):
    """-> dict(xyz f64[P,3], colors f32[P,3] (normalised), segment_ids i64[P], masks bool[P,T],
    segment_connectivity i64[E,2], n_objects)."""
    density = points_per_cell / (VOXEL * VOXEL)
    lo, hi = 0.05, 3.0
    best = None
    for it in range(24):
        s = 0.5 * (lo + hi)
        rng = np.random.default_rng(seed)
        rects, n_obj = _layout(rng, s)
        pts, objs, faces, tiles = _sample(rects, rng, density)
        nv = _count_voxels(pts)
        best = (pts, objs, faces, tiles, n_obj, nv)
        if abs(nv - target_voxels) <= tol * target_voxels:
            break
        if nv > target_voxels:
            hi = s
        else:
            lo = s
    pts, objs, faces, tiles, n_obj, nv = best
    if nv > (1 + tol) * target_voxels:
        # tiny targets (unit tests): the smallest room is still too big -> crop along x
        xs = np.sort(pts[:, 0])
        lo_i, hi_i = 0, pts.shape[0] - 1
        for _ in range(24):
            mid = (lo_i + hi_i) // 2
            keep = pts[:, 0] <= xs[mid]
            nv = _count_voxels(pts[keep])
            if abs(nv - target_voxels) <= tol * target_voxels:
                break
            if nv > target_voxels:
                hi_i = mid
            else:
                lo_i = mid
        pts, objs, faces, tiles = pts[keep], objs[keep], faces[keep], tiles[keep]
    rng = np.random.default_rng(seed + 7919)
    perm = rng.permutation(pts.shape[0])
    pts, objs, faces, tiles = pts[perm], objs[perm], faces[perm], tiles[perm]

    base = rng.uniform(0, 255, (n_obj, 3))
    rgb = np.clip(base[objs] + rng.normal(0, 8, (pts.shape[0], 3)), 0, 255)
    colors = ((rgb - _COLOR_MEAN) / _COLOR_STD).astype(np.float32)

    seg_key = (objs * 8 + faces) * 4096 + tiles
    uniq, segment_ids = np.unique(seg_key, return_inverse=True)
    segment_ids = segment_ids.reshape(-1).astype(np.int64)
    S = uniq.shape[0]

    # tile adjacency inside one face (both directions) -> directed connectivity pairs
    obj_face = uniq // 4096
    tile = uniq % 4096
    tu, tv = tile // 64, tile % 64
    index = {(int(of), int(a), int(b)): i for i, (of, a, b) in enumerate(zip(obj_face, tu, tv))}
    conn = []
    for i, (of, a, b) in enumerate(zip(obj_face, tu, tv)):
        for da, db in ((1, 0), (0, 1)):
            j = index.get((int(of), int(a + da), int(b + db)))
            if j is not None:
                conn.append((i, j))
                conn.append((j, i))
    conn = np.asarray(conn, np.int64).reshape(-1, 2)

    T = min(n_obj - 5, 20)  # furniture only (skip floor + 4 walls), at most 20 pseudo masks
    masks = np.zeros((pts.shape[0], T), bool)
    for t in range(T):
        masks[:, t] = objs == (5 + t)
    return {
        "xyz": pts.astype(np.float64), "colors": colors, "segment_ids": segment_ids, "masks": masks,
        "segment_connectivity": conn, "n_objects": n_obj, "n_segments": S, "n_voxels_estimate": nv,
    }


def make_segment_scene(seed: int, side: int = 25, dims=(384, 96), n_objects: int = 16, noise=(0.3, 1.2)):
    """Config-5 inputs: a side x side grid of oversegmentation segments (side=25 -> 625, the
    "600-segment scene"), `n_objects` compact rectangular objects of distinct sizes on a background cluster,
    per-modality features = cluster centre + per-segment sigma ~ U(noise) noise (dims: DINO-like 384, CSC-like
    96; irregular thresholded graphs like real features rather than block-constant ones), and the 4-neighbour
    directed segment connectivity.  -> (feats list of f32[S,d], conn i64[E,2], label i64[S])."""
    rng = np.random.default_rng(seed)
    S = side * side
    label = np.zeros((side, side), np.int64)
    for k in range(n_objects):
        h, w = 2 + (k % 4), 3 + (k // 3)
        for _ in range(200):
            r, c = int(rng.integers(1, side - h - 1)), int(rng.integers(1, side - w - 1))
            if not label[r - 1:r + h + 1, c - 1:c + w + 1].any():
                label[r:r + h, c:c + w] = k + 1
                break
    label = label.reshape(-1)
    sig = rng.uniform(noise[0], noise[1], size=(S, 1))
    feats = []
    for d in dims:
        cent = rng.normal(size=(n_objects + 1, d))
        feats.append((cent[label] + sig * rng.normal(size=(S, d))).astype(np.float32))
    idx = np.arange(S).reshape(side, side)
    right = np.stack([idx[:, :-1].reshape(-1), idx[:, 1:].reshape(-1)], 1)
    down = np.stack([idx[:-1].reshape(-1), idx[1:].reshape(-1)], 1)
    und = np.concatenate([right, down])
    conn = np.concatenate([und, und[:, ::-1]]).astype(np.int64)
    return feats, conn, label


# ---- 2D -> 3D projection inputs (a voxelised room and camera poses inside it) ----------------------------------
def room_voxels(seed, dims=(40, 36, 28), n_boxes=4, batch=1, origin=(-7, 11, 3)):
    """Voxel shell of a room (floor, ceiling, walls) with a few boxes inside, rows shuffled."""
    rng = np.random.default_rng(seed)
    out = []
    for b in range(batch):
        X, Y, Z = dims
        g = np.zeros((X, Y, Z), bool)
        g[0], g[-1], g[:, 0], g[:, -1], g[:, :, 0], g[:, :, -1] = True, True, True, True, True, True
        for _ in range(n_boxes):
            lo = rng.integers(3, [X - 10, Y - 10, Z - 10])
            sz = rng.integers(3, 8, 3)
            g[lo[0]:lo[0] + sz[0], lo[1]:lo[1] + sz[1], lo[2]:lo[2] + sz[2]] = True
        xyz = np.argwhere(g)
        xyz = xyz[rng.permutation(len(xyz))] + np.asarray(origin) + b
        out.append(np.concatenate([np.full((len(xyz), 1), b), xyz], 1))
    return np.concatenate(out).astype(np.int32)


def look_at(eye, target, up=(0, 0, 1)):
    """Camera-to-world 4x4 (camera z forward, x right, y down)."""
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    z = target - eye
    z /= np.linalg.norm(z)
    x = np.cross(z, up)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, eye
    return m.astype(np.float32)


def camera_views(seed, coords, n_views, dims=(40, 36, 28)):
    rng = np.random.default_rng(seed)
    B = int(coords[-1, 0]) + 1
    views = np.zeros((B, n_views, 4, 4), np.float32)
    for b in range(B):
        lo = coords[coords[:, 0] == b, 1:].min(0)
        for v in range(n_views):
            eye = lo + np.asarray(dims) * rng.uniform(0.35, 0.65, 3)
            tgt = lo + np.asarray(dims) * rng.uniform(0.1, 0.9, 3)
            views[b, v] = look_at(eye, tgt)
    return views


def make_mesh(seed: int, side: int = 60, n_regions: int = 9, colour_noise: float = 0.02, quantise_colours: bool = False):
    """Synthetic triangle mesh for the over-segmentation: a `side` x `side` height field (two triangles
    per cell) made of flat and tilted patches with creases between them, one base colour per patch plus per-vertex
    noise.  `quantise_colours`: colours rounded to 1/16 so that many edges get EQUAL weights (ties, incl. exact zeros).
    -> vertices f32[V,3], faces i32[F,3], colors f32[V,3] in [0,1]."""
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
    cx, cy = rng.uniform(0, side, n_regions), rng.uniform(0, side, n_regions)
    region = np.argmin((u[..., None] - cx) ** 2 + (v[..., None] - cy) ** 2, axis=-1)
    slope = rng.uniform(-0.6, 0.6, (n_regions, 2))
    base = rng.uniform(0, 1.5, n_regions)
    z = base[region] + slope[region, 0] * (u - cx[region]) * 0.05 + slope[region, 1] * (v - cy[region]) * 0.05
    z = z + rng.normal(0, 0.002, z.shape)
    vertices = np.stack([u * 0.05, v * 0.05, z], -1).reshape(-1, 3).astype(np.float32)
    colors = rng.uniform(0.1, 0.9, (n_regions, 3))[region] + rng.normal(0, colour_noise, (side, side, 3))
    colors = np.clip(colors, 0, 1)
    if quantise_colours:
        colors = np.round(colors * 16) / 16
    colors = colors.reshape(-1, 3).astype(np.float32)
    idx = (u * side + v)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[:-1, 1:], idx[1:, 1:]
    faces = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([b, d, c], -1).reshape(-1, 3)], 0)
    faces = faces[rng.permutation(faces.shape[0])].astype(np.int32)      # face order matters (running normal blend)
    return vertices, faces, colors
