"""2D -> 3D feature projection: oracle known answers on the CPU, HIP kernels against the
oracle on the GPU (bit-exact: hit rows, counts, and — because both sum in ascending pixel order — features)."""
import math

import numpy as np
import pytest
import torch

from oracle import project_ref as PR
from unscene3d_amd.synthetic import camera_views as cameras, look_at, room_voxels as room

F = np.float32


def intrinsics(B, W, H):
    return np.tile(np.array([[W * 0.9, W * 0.9, (W - 1) / 2 + 0.25, (H - 1) / 2 - 0.4]], F), (B, 1))


DMIN, DMAX, INC = 0.1 / 0.02, 0.9 / 0.02, 0.01


def scalar_first_hit(occ_b, view, intr, x, y):
    """Independent scalar restatement of one ray (np.float32 scalars, python loop)."""
    m = view
    fx, fy, mx, my = (F(v) for v in intr)
    depth = F(1.0) * (F(DMAX) - F(DMIN)) + F(DMIN)
    a = [depth * ((F(x) - mx) / fx), depth * ((F(y) - my) / fy), depth]
    inv = F(1.0) / np.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])
    a = [v * inv for v in a]
    w = [m[i, 0] * a[0] + m[i, 1] * a[1] + m[i, 2] * a[2] + m[i, 3] * F(0) for i in range(3)]
    inv = F(1.0) / np.sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2])
    d = [v * inv for v in w]
    t, t1 = (F(1.0) / a[2]) * F(DMIN), (F(1.0) / a[2]) * F(DMAX)
    dz, dy, dx = occ_b.shape
    while t < t1:
        p = [m[i, 3] + t * d[i] for i in range(3)]
        q = [int(v + F(0.5) * F(np.sign(v))) for v in p]
        if 0 <= q[0] < dx and 0 <= q[1] < dy and 0 <= q[2] < dz and occ_b[q[2], q[1], q[0]] != 0:
            return int(occ_b[q[2], q[1], q[0]])
        t = t + F(INC)
    return -1


# ----------------------------------------------------------------------------------------------------------- CPU
def test_oracle_axis_aligned_known_answers():
    # a wall at z = 20 (all x, y in a 21 x 21 patch), camera on the axis through its centre, looking along +z
    xs, ys = np.meshgrid(np.arange(21), np.arange(21), indexing="ij")
    wall = np.stack([np.zeros(441, int), xs.ravel(), ys.ravel(), np.full(441, 20)], 1)
    blocker = np.array([[0, 10, 10, 12]])          # row 0, right in front of the camera: must be ignored
    coords = np.concatenate([blocker, wall]).astype(np.int32)
    occ, shifts = PR.dense_occupancy(coords)
    assert occ.shape == (1, 9, 21, 21) and tuple(shifts[0]) == (0, 0, 12)
    view = np.eye(4, dtype=F)
    view[:3, 3] = [10, 10, 2]
    views = PR.shift_views(view[None, None], shifts)
    W = H = 9
    intr = np.array([[30.0, 30.0, 4.0, 4.0]], F)
    hit = PR.first_hit(occ, views, intr, W, H, 5.0, 45.0, 0.01)[0, 0]
    row_of = {tuple(c[1:]): i for i, c in enumerate(coords)}
    assert hit[4, 4] == row_of[(10, 10, 20)]           # centre ray passes through voxel row 0 and hits the wall
    assert (hit >= 1).all()
    # the wall is hit where the ray crosses z = 19.5: x = 10 + (ux - 4)/30 * 17.5
    for uy in range(H):
        for ux in range(W):
            ex, ey = 10 + (ux - 4) / 30 * 17.5, 10 + (uy - 4) / 30 * 17.5
            hx, hy, hz = coords[hit[uy, ux], 1:]
            assert hz == 20 and abs(hx - ex) <= 0.51 and abs(hy - ey) <= 0.51
    # no voxel within [depth_min, depth_max]: no hit
    assert (PR.first_hit(occ, views, intr, W, H, 1.0, 4.0, 0.01) == -1).all()


def test_oracle_vector_march_equals_scalar_loop():
    coords = room(3)
    occ, shifts = PR.dense_occupancy(coords)
    views = PR.shift_views(cameras(5, coords, 2), shifts)
    W, H = 6, 5
    intr = intrinsics(1, W, H)
    hit = PR.first_hit(occ, views, intr, W, H, DMIN, DMAX, INC)
    assert (hit >= 0).mean() > 0.5
    for v in range(2):
        for (x, y) in [(0, 0), (5, 4), (2, 3), (4, 1)]:
            assert hit[0, v, y, x] == scalar_first_hit(occ[0], views[0, v], intr[0], x, y)


def test_oracle_reduce_fuse_and_predictions():
    rng = np.random.default_rng(0)
    hit = np.array([[[[3, -1, 3], [1, 3, -1]]]], np.int32)
    feats = rng.standard_normal((1, 1, 2, 3, 4)).astype(F)
    proj, num = PR.project_features(feats, hit, 5)
    assert num.tolist() == [0, 1, 0, 3, 0]
    f = feats.reshape(-1, 4)
    np.testing.assert_array_equal(proj[3], ((f[0] + f[2]) + f[4]) / (F(3) + F(10e-5)))
    np.testing.assert_array_equal(proj[1], f[3] / (F(1) + F(10e-5)))
    assert not proj[[0, 2, 4]].any()
    scene = rng.standard_normal((5, 4)).astype(F)
    fused = PR.fuse_frame(scene, proj, num)
    np.testing.assert_array_equal(fused[[0, 2, 4]], scene[[0, 2, 4]])
    np.testing.assert_array_equal(fused[3], (scene[3] + proj[3]) / F(2))
    preds = rng.integers(0, 400, (1, 1, 2, 3, 1)).astype(np.int32)
    out = PR.project_predictions(preds, hit, 5, 255)
    p = preds.reshape(-1)
    assert out[:, 0].tolist() == [255, max(255, p[3]), 255, max(255, p[0], p[2], p[4]), 255]


# ----------------------------------------------------------------------------------------------------------- GPU
def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


@pytest.mark.gpu
@pytest.mark.parametrize("batch,n_views,W,H", [(1, 1, 24, 18), (2, 3, 19, 13)])
def test_first_hit_hash_and_dense_match_oracle(device, batch, n_views, W, H):
    from unscene3d_amd import ops
    from unscene3d_amd import project_features_cuda as P

    coords = room(11 + batch, batch=batch)
    occ, shifts = PR.dense_occupancy(coords)
    views = PR.shift_views(cameras(7, coords, n_views), shifts)
    intr = intrinsics(batch, W, H)
    ref = PR.first_hit(occ, views, intr, W, H, DMIN, DMAX, INC)
    assert 0.3 < (ref >= 0).mean() <= 1.0

    n = coords.shape[0]
    hit_d, seg_d = P.raycast_first_hit_dense(_dev(occ, device), _dev(views, device), _dev(intr, device), H, W, DMIN,
                                             DMAX, INC, n)
    assert np.array_equal(hit_d.cpu().numpy(), ref)
    cmap, _, _ = ops.coordmap_build(_dev(coords, device))
    assert cmap.n == n
    hit_h, seg_h = P.raycast_first_hit_map(cmap, _dev(shifts.astype(np.int32), device), _dev(views, device),
                                           _dev(intr, device), H, W, DMIN, DMAX, INC)
    assert np.array_equal(hit_h.cpu().numpy(), ref)
    sh_d = _dev(shifts.astype(np.int32), device)
    hit_b, _ = P.raycast_first_hit_map(cmap, sh_d, _dev(views, device), _dev(intr, device), H, W, DMIN, DMAX, INC,
                                       bricks=P.brick_mask(cmap, sh_d))       # free-space filter: same hits
    assert np.array_equal(hit_b.cpu().numpy(), ref)
    exp_seg = np.where(ref.reshape(-1) >= 0, ref.reshape(-1), n).astype(np.int64)
    assert np.array_equal(seg_d.cpu().numpy(), exp_seg) and np.array_equal(seg_h.cpu().numpy(), exp_seg)


@pytest.mark.gpu
def test_projection_module_operator_fusion_and_predictions(device):
    from types import SimpleNamespace

    from unscene3d_amd import project_features_cuda as P

    batch, n_views, W, H, C = 2, 2, 20, 14, 70
    coords = room(21, batch=batch)
    n = coords.shape[0]
    occ, shifts = PR.dense_occupancy(coords)
    raw_views = cameras(9, coords, n_views)
    views = PR.shift_views(raw_views, shifts)
    intr = intrinsics(batch, W, H)
    rng = np.random.default_rng(2)
    feats = rng.standard_normal((batch, n_views, H, W, C)).astype(F)
    ref_hit = PR.first_hit(occ, views, intr, W, H, DMIN, DMAX, INC)
    ref_proj, ref_num = PR.project_features(feats, ref_hit, n)

    # module: same constructor / forward as the reference's Project2DFeaturesCUDA (unshifted views go in)
    proj = P.Project2DFeaturesCUDA(width=W, height=H, voxel_size=0.02, depth_min=DMIN * 0.02, depth_max=DMAX * 0.02)
    assert math.isclose(proj.ray_increment, INC) and math.isclose(proj.depth_max, DMAX)
    c_d = _dev(coords, device)
    out, num = proj(_dev(feats, device), c_d, _dev(raw_views, device), _dev(intr, device))
    assert out.shape == (n, C) and num.dtype == torch.int32
    assert np.array_equal(num.cpu().numpy(), ref_num)
    assert np.array_equal(out.cpu().numpy(), ref_proj)              # same summation order: bit-exact

    # the extension's operator: accumulates raw sums / counts into caller tensors
    acc = torch.ones((n, C), device=device)
    cnt = torch.full((n,), 2, dtype=torch.int32, device=device)
    opts = torch.tensor([W, H, DMIN, DMAX, INC], dtype=torch.float32)
    P.project_features_cuda(_dev(feats, device), _dev(occ, device), _dev(views, device), _dev(intr, device), opts, cnt,
                            acc, torch.BoolTensor([False]))
    assert np.array_equal(cnt.cpu().numpy(), ref_num + 2)
    raw = np.zeros((n, C), F)
    flat = ref_hit.reshape(-1)
    pix = np.nonzero(flat >= 0)[0]
    np.add.at(raw, flat[pix], feats.reshape(-1, C)[pix])
    np.testing.assert_allclose(acc.cpu().numpy(), raw + 1, rtol=1e-6, atol=1e-6)

    # running mean over frames, one view at a time (unscene3d_pseudo_main.py:303-313), batch 0 only like the caller
    c0 = coords[coords[:, 0] == 0]
    n0 = c0.shape[0]
    occ0, sh0 = PR.dense_occupancy(c0)
    scene_ref = np.zeros((n0, C), F)
    scene = torch.zeros((n0, C), device=device)
    c0_d = _dev(c0, device)
    for v in range(n_views):
        vv = raw_views[:1, v:v + 1]
        h = PR.first_hit(occ0, PR.shift_views(vv, sh0), intr[:1], W, H, DMIN, DMAX, INC)
        pr, nm = PR.project_features(feats[:1, v:v + 1], h, n0)
        scene_ref = PR.fuse_frame(scene_ref, pr, nm)
        num_v, _ = proj.fuse_frame(scene, _dev(feats[:1, v:v + 1], device), c0_d, _dev(vv, device),
                                   _dev(intr[:1], device))
        assert np.array_equal(num_v.cpu().numpy(), nm)
    assert np.array_equal(scene.cpu().numpy(), scene_ref)

    # prediction mode: integer max against the ignore label, counts stay zero (the kernel never touches them)
    cfg = SimpleNamespace(data=SimpleNamespace(ignore_label=255))
    projp = P.Project2DFeaturesCUDA(W, H, 0.02, cfg, depth_min=DMIN * 0.02, depth_max=DMAX * 0.02)
    preds = rng.integers(0, 600, (batch, n_views, H, W, 1)).astype(np.int32)
    lab, num0 = projp(_dev(preds, device), c_d, _dev(raw_views, device), _dev(intr, device), pred_mode=True)
    assert lab.dtype == torch.int64 and lab.shape == (n,) and int(num0.sum()) == 0
    assert np.array_equal(lab.cpu().numpy(), PR.project_predictions(preds, ref_hit, n, 255).reshape(-1))


@pytest.mark.gpu
def test_unproject_depth_images(device):
    from unscene3d_amd import project_features_cuda as P

    rng = np.random.default_rng(4)
    V, H, W = 3, 11, 17
    depth = rng.uniform(-0.5, 4.0, (V, H, W)).astype(F)
    coords = room(1)
    views = cameras(3, coords, V)[0] * F(0.02)
    intr = np.tile(intrinsics(1, W, H), (V, 1))
    cloud = torch.zeros((V * H * W, 5), device=device)
    P.unproject_depth_images(_dev(depth, device), _dev(views, device), _dev(intr, device), cloud)
    assert np.array_equal(cloud.cpu().numpy(), PR.unproject_depth(depth, views, intr))


@pytest.mark.gpu
def test_full_size_frame_properties(device):
    """BASELINE-size frame (192 x 256 rays, 384 channels) on a ~150 k-voxel scene, checked through properties
    that need no oracle run: hash == dense occupancy, counts == hits, every hit voxel lies on its pixel's ray
    inside the depth range, and no occupied voxel closer to the camera was skipped (sampled rays)."""
    from unscene3d_amd import ops
    from unscene3d_amd import project_features_cuda as P

    dims = (150, 170, 130)
    coords = room(5, dims=dims, n_boxes=30)
    n = coords.shape[0]
    assert n > 100_000
    occ, shifts = PR.dense_occupancy(coords)
    H, W, C = 192, 256, 384
    raw_views = cameras(6, coords, 1, dims=dims)
    views = PR.shift_views(raw_views, shifts)
    intr = np.array([[W * 0.9, W * 0.9, (W - 1) / 2, (H - 1) / 2]], F)
    dmin, dmax = 0.1 / 0.02, 4.0 / 0.02
    cmap, _, _ = ops.coordmap_build(_dev(coords, device))
    hit_h, seg = P.raycast_first_hit_map(cmap, _dev(shifts.astype(np.int32), device), _dev(views, device),
                                         _dev(intr, device), H, W, dmin, dmax, INC)
    hit_d, _ = P.raycast_first_hit_dense(_dev(occ, device), _dev(views, device), _dev(intr, device), H, W, dmin, dmax,
                                         INC, n)
    assert torch.equal(hit_h, hit_d)
    sh_d = _dev(shifts.astype(np.int32), device)
    hit_b, _ = P.raycast_first_hit_map(cmap, sh_d, _dev(views, device), _dev(intr, device), H, W, dmin, dmax, INC,
                                       bricks=P.brick_mask(cmap, sh_d))
    assert torch.equal(hit_b, hit_d)
    hit = hit_h.cpu().numpy()[0, 0]
    assert (hit >= 0).mean() > 0.9

    feats = torch.randn((1, 1, H, W, C), device=device)
    out = torch.empty((n, C), device=device)
    num = torch.empty(n, dtype=torch.int32, device=device)
    P.project_reduce(feats.view(-1, C), seg, n, out, num)
    assert int(num.sum()) == int((hit >= 0).sum())
    # sum over voxels of count * mean == sum over hit pixels (linearity), in f64 on the host
    tot = (out.double() * (num.double()[:, None] + 10e-5)).sum(0).cpu().numpy()
    exp = feats.view(-1, C)[torch.from_numpy(hit.reshape(-1) >= 0).to(device)].double().sum(0).cpu().numpy()
    np.testing.assert_allclose(tot, exp, rtol=1e-4, atol=1e-2)

    # geometry of the hits
    cam, d, t0, t1 = PR.ray_setup(views[0, 0], intr[0], W, H, dmin, dmax)
    local = (coords[:, 1:] - shifts[0]).astype(np.float64)
    ys, xs = np.nonzero(hit >= 0)
    centre = local[hit[ys, xs]]
    dd = d[ys, xs].astype(np.float64)
    rel = centre - cam.astype(np.float64)
    t = (rel * dd).sum(1)
    perp = np.linalg.norm(rel - t[:, None] * dd, axis=1)
    assert (perp <= math.sqrt(3) / 2 + 1e-3).all()
    assert (t >= t0[ys, xs] - 1.0).all() and (t <= t1[ys, xs] + 1.0).all()
    rng = np.random.default_rng(0)
    for k in rng.choice(len(ys), 40, replace=False):     # nothing occupied before the hit (coarser f64 march)
        for s in np.arange(float(t0[ys[k], xs[k]]), t[k] - 1.0, 0.05):
            q = np.floor(cam + s * dd[k] + 0.5).astype(int)
            if (q >= 0).all() and (q < np.array(dims)).all():
                v = occ[0, q[2], q[1], q[0]]
                assert v == 0 or v == hit[ys[k], xs[k]]


@pytest.mark.gpu
def test_frame_that_sees_nothing(device):
    """Depth range in front of every wall: no pixel hits, features and counts stay zero, the fused update leaves the
    scene untouched."""
    from unscene3d_amd import project_features_cuda as P

    coords = room(3)
    n = coords.shape[0]
    W, H, C = 12, 9, 5
    proj = P.Project2DFeaturesCUDA(W, H, 0.02, depth_min=0.02, depth_max=0.05)     # 1 .. 2.5 voxels
    c_d = _dev(coords, device)
    lo = coords[:, 1:].min(0)
    eye = lo + np.array([20.0, 18.0, 14.0])
    views = look_at(eye, eye + np.array([1.0, 0.2, 0.1]))[None, None]
    feats = torch.randn((1, 1, H, W, C), device=device)
    out, num = proj(feats, c_d, _dev(views, device), _dev(intrinsics(1, W, H), device))
    assert int(num.sum()) == 0 and not bool(out.any())
    scene = torch.randn((n, C), device=device)
    before = scene.clone()
    num2, _ = proj.fuse_frame(scene, feats, c_d, _dev(views, device), _dev(intrinsics(1, W, H), device))
    assert int(num2.sum()) == 0 and torch.equal(scene, before)
