"""Known-answer tests that pin oracle/sparse_ref.py (the ME restatement has no
runnable reference here — SURVEY.md §8c): sparse conv == dense conv3d restricted
to occupied sites, transposed conv == conv_transpose3d on the cached fine sites,
unique / neighbour search == brute force."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import sparse_ref as R


def _random_coords(rng, n, extent, batch=1):
    c = rng.integers(0, extent, size=(n, 3))
    b = rng.integers(0, batch, size=(n, 1))
    return np.concatenate([b, c], 1).astype(np.int32)


def test_unique_first_occurrence_matches_bruteforce():
    rng = np.random.default_rng(0)
    c = _random_coords(rng, 500, 6, batch=2)
    u, inv, oc = R.coordmap_build(c)
    seen, exp_u, exp_inv = {}, [], []
    for i, row in enumerate(map(tuple, c)):
        if row not in seen:
            seen[row] = len(exp_u)
            exp_u.append(i)
        exp_inv.append(seen[row])
    assert u.tolist() == exp_u
    assert inv.tolist() == exp_inv
    assert np.array_equal(oc, c[u])
    assert np.all(np.diff(u) > 0)


def test_negative_coordinates_floor_and_quantise():
    xyz = np.array([[-0.001, 0.0199, 0.02], [-0.02, -0.0201, 0.0401]])
    assert R.voxel_floor(xyz, 0.02).tolist() == [[-1, 0, 1], [-1, -2, 2]]
    c = np.array([[0, -1, -2, -3], [0, 3, 2, 1], [0, -4, 0, 4]], np.int32)
    q = R.quantize_coords(c, 2)
    assert q.tolist() == [[0, -2, -2, -4], [0, 2, 2, 0], [0, -4, 0, 4]]


def test_kernel_map_cube_bruteforce():
    rng = np.random.default_rng(1)
    c = R.coordmap_build(_random_coords(rng, 300, 7))[2]
    nbr = R.kernel_map_cube(c, 1)
    lut = {tuple(r): i for i, r in enumerate(c.tolist())}
    for k, (dx, dy, dz) in enumerate(R.cube_offsets(3)):
        for o, (b, x, y, z) in enumerate(c.tolist()):
            assert nbr[k, o] == lut.get((b, x + dx, y + dy, z + dz), -1)
    assert np.array_equal(nbr[13], np.arange(len(c)))
    # symmetric map: nbr[k][o] = i  <=>  nbr[26-k][i] = o
    for k in range(27):
        o = np.nonzero(nbr[k] >= 0)[0]
        assert np.array_equal(nbr[26 - k][nbr[k][o]], o)


def _densify(coords, feats, extent, ts=1):
    g = torch.zeros(feats.shape[1], extent, extent, extent, dtype=feats.dtype)
    idx = torch.as_tensor(coords[:, 1:] // ts, dtype=torch.long)
    g[:, idx[:, 2], idx[:, 1], idx[:, 0]] = feats.T   # grid[c, z, y, x]
    return g


def test_sparse_conv_equals_dense_conv3d():
    rng = np.random.default_rng(2)
    E = 9
    c = R.coordmap_build(_random_coords(rng, 250, E))[2]
    n, cin, cout = len(c), 5, 7
    feats = torch.randn(n, cin, dtype=torch.float64)
    W = torch.randn(27, cin, cout, dtype=torch.float64)
    out = R.conv_gather(feats, W, R.kernel_map_cube(c, 1), n)
    # dense: weight[co, ci, kz, ky, kx] with k = kx + 3 ky + 9 kz (x fastest)
    Wd = W.reshape(3, 3, 3, cin, cout).permute(4, 3, 0, 1, 2).contiguous()
    dense = F.conv3d(_densify(c, feats, E)[None], Wd, padding=1)[0]
    idx = torch.as_tensor(c[:, 1:], dtype=torch.long)
    exp = dense[:, idx[:, 2], idx[:, 1], idx[:, 0]].T
    assert torch.allclose(out, exp, atol=1e-10)


def test_strided_conv_and_transpose_equal_dense():
    rng = np.random.default_rng(3)
    E = 8
    c = R.coordmap_build(_random_coords(rng, 200, E))[2]
    n, cin, cout = len(c), 4, 6
    _, parent, cc = R.coordmap_build(c, 2)
    nbr2, kidx = R.kernel_map_down2(c, 1, parent, cc)
    feats = torch.randn(n, cin, dtype=torch.float64)
    W = torch.randn(8, cin, cout, dtype=torch.float64)
    out = R.conv_gather(feats, W, nbr2, len(cc))
    Wd = W.reshape(2, 2, 2, cin, cout).permute(4, 3, 0, 1, 2).contiguous()   # [co,ci,kz,ky,kx]
    dense = F.conv3d(_densify(c, feats, E)[None], Wd, stride=2)[0]
    ci = torch.as_tensor(cc[:, 1:] // 2, dtype=torch.long)
    assert torch.allclose(out, dense[:, ci[:, 2], ci[:, 1], ci[:, 0]].T, atol=1e-10)
    # coarse coordinates are the distinct floor(c/2)*2 in first-occurrence order
    assert np.array_equal(cc, R.quantize_coords(c, 2)[R.coordmap_build(c, 2)[0]])
    # transposed conv back onto the fine sites
    fc = torch.randn(len(cc), cout, dtype=torch.float64)
    Wt = torch.randn(8, cout, cin, dtype=torch.float64)
    up = R.conv_transpose_up2(fc, Wt, parent, kidx, n)
    Wtd = Wt.reshape(2, 2, 2, cout, cin).permute(3, 4, 0, 1, 2).contiguous()  # [cin_t=cout, cout_t=cin, kz,ky,kx]
    dense_up = F.conv_transpose3d(_densify(cc, fc, E // 2, ts=2)[None], Wtd, stride=2)[0]
    fi = torch.as_tensor(c[:, 1:], dtype=torch.long)
    assert torch.allclose(up, dense_up[:, fi[:, 2], fi[:, 1], fi[:, 0]].T, atol=1e-10)


def test_rulebook_compact_matches_table():
    rng = np.random.default_rng(4)
    c = R.coordmap_build(_random_coords(rng, 120, 5))[2]
    nbr = R.kernel_map_cube(c, 1)
    i, o, koff = R.rulebook_compact(nbr)
    assert koff[-1] == (nbr >= 0).sum()
    for k in range(27):
        sl = slice(koff[k], koff[k + 1])
        assert np.array_equal(o[sl], np.nonzero(nbr[k] >= 0)[0])
        assert np.array_equal(i[sl], nbr[k][o[sl]])


def test_avgpool_and_scatter_mean_and_bn():
    rng = np.random.default_rng(5)
    c = R.coordmap_build(_random_coords(rng, 150, 6))[2]
    _, parent, cc = R.coordmap_build(c, 2)
    nbr2, _ = R.kernel_map_down2(c, 1, parent, cc)
    f = torch.randn(len(c), 3, dtype=torch.float64)
    pooled = R.avgpool_down2(f, nbr2)
    exp = R.scatter_mean(f, torch.as_tensor(parent), len(cc))
    assert torch.allclose(pooled, exp, atol=1e-12)
    x = torch.randn(64, 5)
    y, mean, var = R.batch_norm_train(x, torch.ones(5), torch.zeros(5))
    exp = F.batch_norm(x, None, None, torch.ones(5), torch.zeros(5), training=True, eps=1e-5)
    assert torch.allclose(y, exp, atol=1e-5)
