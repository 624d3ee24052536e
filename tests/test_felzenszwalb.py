"""Felzenszwalb mesh over-segmentation (reference utils/cpp_utils/segmentator.cpp:17-154).

Chain of evidence: the reference's own extension module, compiled from /root/reference by `make -C oracle ref`
(oracle/_ref/, build container only), produced tests/golden/felz.npz and pins oracle/felz_oracle.cpp (labels and
connectivity identical, also with thousands of tied weights).  The product path (device normals / weights / stable
sort + usc_felz_merge_host) must give bit-equal normals and weights and the same PARTITION and segment adjacency; its
label numbers are the oracle's in stable-sort mode (equal weights in face order)."""
import os

import numpy as np
import pytest

from oracle import felz_ref as FR

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "felz.npz")
CASES = ("distinct", "ties")


def _case(z, name):
    return tuple(z[f"{name}/{k}"] for k in ("vertices", "faces", "colors", "labels", "connectivity", "normals", "weights"))


def _canon(labels, pairs):
    """labels renumbered by first occurrence + the pair set under that renumbering (partition-level comparison)."""
    _, first, inv = np.unique(labels, return_index=True, return_inverse=True)
    rank = np.empty(len(first), np.int64)
    rank[np.argsort(first)] = np.arange(len(first))
    return rank[inv.reshape(-1)], {(int(rank[a]), int(rank[b])) for a, b in pairs}


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    v, f, c, labels, conn, normals, weights = _case(np.load(GOLD), name)
    ol, oc, det = FR.segment_mesh(v, f, c, 0.005, 20, stable=False, details=True)
    assert np.array_equal(ol, labels) and np.array_equal(oc, conn)                  # the reference's std::sort order
    assert np.array_equal(det["normals"].view(np.int32), normals.view(np.int32))
    assert np.array_equal(det["weights"].view(np.int32), weights.view(np.int32))
    # canonical (stable) edge order: same partition and adjacency, only representatives / label numbers may move
    sl, sc = FR.segment_mesh(v, f, c, 0.005, 20, stable=True)
    a, b = _canon(labels, conn), _canon(sl, sc)
    assert np.array_equal(a[0], b[0]) and a[1] == b[1]


def test_oracle_equals_reference_build():
    """Only where oracle/_ref holds the reference module (build container, or shipped prebuilt to the GPU box)."""
    ref = FR.reference_module()
    if ref is None:
        pytest.skip("oracle/_ref not built (make -C oracle ref needs /root/reference)")
    from unscene3d_amd.synthetic import make_mesh
    for seed, side, q in ((11, 40, False), (12, 40, True), (13, 120, False), (14, 120, True)):
        v, f, c = make_mesh(seed, side=side, quantise_colours=q)
        for min_verts in (20, 50):
            rl, rc = ref.segment_mesh(v, f, c, 0.005, min_verts)
            ol, oc = FR.segment_mesh(v, f, c, 0.005, min_verts, stable=False)
            assert np.array_equal(rl, ol) and np.array_equal(np.asarray(rc), oc), (seed, min_verts)


@pytest.mark.parametrize("name", CASES)
def test_host_merge_equals_oracle(name):
    """usc_felz_merge_host (the product's sequential part; needs no GPU) on the golden weights in stable order."""
    from unscene3d_amd import felzenszwalb_cpp as FZ

    v, f, c, labels, conn, normals, weights = _case(np.load(GOLD), name)
    ea = np.stack([f[:, 0], f[:, 0], f[:, 2]], 1).reshape(-1)
    eb = np.stack([f[:, 1], f[:, 2], f[:, 1]], 1).reshape(-1)
    order = np.argsort(weights, kind="stable")
    comps = FZ.merge_host(ea[order], eb[order], weights[order], v.shape[0], 0.005, 20)
    got = FZ.relabel(comps, ea[order], eb[order])
    exp = FR.segment_mesh(v, f, c, 0.005, 20, stable=True)
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])
    with pytest.raises(RuntimeError):
        FZ.merge_host(np.array([0, 7], np.int32), np.array([1, 2], np.int32), np.zeros(2, np.float32), 3, 0.005, 20)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_path_matches_reference_golden(device, name):
    import torch

    from unscene3d_amd import felzenszwalb_cpp as FZ

    v, f, c, labels, conn, normals, weights = _case(np.load(GOLD), name)
    dv, df, dc = (torch.from_numpy(x).to(device) for x in (v, f, c))
    ea, eb, w, n = FZ.edge_weights(dv, df, dc)
    assert np.array_equal(n.cpu().numpy().view(np.int32), normals.view(np.int32))          # bit for bit
    assert np.array_equal(w.cpu().numpy().view(np.int32), weights.view(np.int32))
    assert np.array_equal(ea.cpu().numpy(), np.stack([f[:, 0], f[:, 0], f[:, 2]], 1).reshape(-1))
    assert np.array_equal(eb.cpu().numpy(), np.stack([f[:, 1], f[:, 2], f[:, 1]], 1).reshape(-1))
    got = FZ.segment_mesh(v, f, c, 0.005, 20, device=device)
    exp = FR.segment_mesh(v, f, c, 0.005, 20, stable=True)
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])            # the oracle's stable mode, exactly
    a, b = _canon(labels, conn), _canon(*got)
    assert np.array_equal(a[0], b[0]) and a[1] == b[1]                                   # the reference's partition


@pytest.mark.gpu
def test_device_path_on_a_scene_sized_mesh(device):
    """160 k vertices / 320 k faces (a ScanNet scene's order of magnitude): device path == oracle (stable), and
    structural properties: every segment has at least segMinVerts vertices or is a whole connected component."""
    from unscene3d_amd import felzenszwalb_cpp as FZ
    from unscene3d_amd.synthetic import make_mesh

    v, f, c = make_mesh(21, side=400, n_regions=60)
    got = FZ.segment_mesh(v, f, c, 0.005, 20, device=device)
    exp = FR.segment_mesh(v, f, c, 0.005, 20, stable=True)
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])
    sizes = np.bincount(got[0])
    assert sizes.min() >= 20 and got[0].max() + 1 == len(sizes)
