"""Self-train mask merge + validation-mode scene reader against golden vectors recorded from
the reference's own `SemanticSegmentationFreeDataset.load_self_train_masks` / `__getitem__`
(tests/golden/dataset.npz, generator: make_golden.py dataset).  CPU: torch CPU tensors and a KD-tree stand-in for
the device 1-NN (host logic); GPU: the product path."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset.npz")


def _kdtree_knn1(monkeypatch):
    from scipy.spatial import KDTree

    from unscene3d_amd import ops

    def knn1(q, r):
        d, i = KDTree(r.double().numpy()).query(q.double().numpy(), k=1)
        return torch.from_numpy(d ** 2), torch.from_numpy(i)

    monkeypatch.setattr(ops, "knn1", knn1)


def _merge_cases(z, device):
    from unscene3d_amd.datasets.freemask import load_self_train_masks

    pts, free, cloud = z["points"], z["freemasks"], z["st_cloud"]
    st = np.unpackbits(z["st_masks"], axis=1)[:, :8].astype(bool)
    for nsd, drop in ((5, False), (2, False), (8, True)):
        got = load_self_train_masks(pts, free, cloud, st, nsd, drop_original=drop, device=device)
        ref = z[f"merge/{nsd}_{int(drop)}"]
        assert got.dtype == np.float32 and got.shape == ref.shape
        assert np.array_equal(got, ref)
    assert ref.shape[1] > 0


def _reader_case(z, device, tmp_path):
    from unscene3d_amd.datasets.freemask import FreeMaskSceneReader
    from unscene3d_amd.trainer.postprocess import save_for_freemask

    d = tmp_path / "scans" / "scene0001_00"
    d.mkdir(parents=True)
    np.save(d / "0001_00.npy", z["points"])
    np.save(d / "0001_00_freemasks.npy", z["freemasks"])
    same = np.unpackbits(z["st_masks_same"], axis=1)[:, :6].astype(bool)
    save_for_freemask(str(tmp_path), "scene0001_00", z["points"][:, :3], same)       # the export side writes ...
    reader = FreeMaskSceneReader([{"filepath": str(d / "0001_00.npy"),
                                   "raw_filepath": "raw/scene0001_00/scene0001_00_vh_clean_2.ply"}],
                                 add_raw_coordinates=True, load_self_train_data=True,
                                 self_train_data_dir=str(tmp_path), device=device)     # ... what the reader takes
    item = reader[0]
    assert item[3] == "scene0001_00" and item[7] == 0 and item[8] == []
    assert np.array_equal(item[0], z["item/coordinates"])
    np.testing.assert_allclose(item[1], z["item/features"], rtol=1e-6, atol=1e-6)
    assert item[2].dtype == np.int32 and np.array_equal(item[2], z["item/freemasks"])
    for got, key in ((item[4], "raw_color"), (item[5], "raw_normals"), (item[6], "raw_coordinates")):
        assert np.array_equal(got, z[f"item/{key}"])


def test_self_train_merge_host_logic(monkeypatch):
    _kdtree_knn1(monkeypatch)
    _merge_cases(np.load(GOLD), "cpu")


def test_scene_reader_host_logic(monkeypatch, tmp_path):
    _kdtree_knn1(monkeypatch)
    _reader_case(np.load(GOLD), "cpu", tmp_path)


@pytest.mark.gpu
def test_self_train_merge_device(device):
    _merge_cases(np.load(GOLD), device)


@pytest.mark.gpu
def test_scene_reader_device_feeds_the_collate(device, tmp_path):
    from unscene3d_amd.datasets.freemask import FreeMaskSceneReader
    from unscene3d_amd.datasets.utils import FreeMaskVoxelizeCollate

    z = np.load(GOLD)
    _reader_case(z, device, tmp_path)
    reader = FreeMaskSceneReader([{"filepath": str(tmp_path / "scans" / "scene0001_00" / "0001_00.npy"),
                                   "raw_filepath": "raw/scene0001_00/scene0001_00_vh_clean_2.ply"}], device=device)
    data, target, names = FreeMaskVoxelizeCollate(voxel_size=0.02, mode="validation", device=device)([reader[0]])
    assert names == ["scene0001_00"] and len(target) == 1
    n_vox = data.coordinates.shape[0]
    assert 0 < n_vox <= 4000 and data.features.shape == (n_vox, 6)
    assert target[0]["point2segment"].shape[0] == n_vox and target[0]["segment_mask"].shape[0] >= 1


def test_reader_without_export_files_keeps_the_scene_masks(monkeypatch, tmp_path, capsys):
    """No `freemasks/scene…` files for a scene: the reference prints a notice and carries on with the scene's own
    pseudo masks (freemask_semseg.py:233-235)."""
    from unscene3d_amd.datasets.freemask import FreeMaskSceneReader

    _kdtree_knn1(monkeypatch)
    z = np.load(GOLD)
    d = tmp_path / "scans" / "scene0001_00"
    d.mkdir(parents=True)
    np.save(d / "0001_00.npy", z["points"])
    np.save(d / "0001_00_freemasks.npy", z["freemasks"])
    entry = {"filepath": str(d / "0001_00.npy"), "raw_filepath": "raw/scene0001_00/scene0001_00_vh_clean_2.ply"}
    plain = FreeMaskSceneReader([entry], device="cpu")[0]
    missing = FreeMaskSceneReader([entry], load_self_train_data=True, self_train_data_dir=str(tmp_path), device="cpu")[0]
    assert "Could not load self training data" in capsys.readouterr().out
    assert np.array_equal(plain[2], missing[2])


class _IdentityVolume:
    transforms = [None]

    def __call__(self, points, normals, features, labels):
        return {"points": points, "normals": normals, "features": features, "labels": labels}


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4, 11])
def test_train_mode_reader_matches_reference(device, tmp_path, seed):
    """TRAIN mode `__getitem__` (freemask_semseg.py:333-437) on the device against the reference run with the same
    seeds (tests/golden/dataset.npz train{seed}: identity stand-ins for the two third-party pipelines): centring +
    random shift, axis flips, elastic-distortion gate and the two distortions, colour drop (seed 4), normalisation,
    and — `rng_after` — that numpy's and python's generators were advanced exactly as far as by the reference."""
    import random as pyrandom

    from unscene3d_amd.datasets.freemask import FreeMaskSceneReader
    from unscene3d_amd.trainer.postprocess import save_for_freemask

    z = np.load(GOLD)
    d = tmp_path / "scans" / "scene0001_00"
    d.mkdir(parents=True)
    np.save(d / "0001_00.npy", z["points"])
    np.save(d / "0001_00_freemasks.npy", z["freemasks"])
    same = np.unpackbits(z["st_masks_same"], axis=1)[:, :6].astype(bool)
    save_for_freemask(str(tmp_path), "scene0001_00", z["points"][:, :3], same)
    reader = FreeMaskSceneReader([{"filepath": str(d / "0001_00.npy"),
                                   "raw_filepath": "raw/scene0001_00/scene0001_00_vh_clean_2.ply"}],
                                 add_raw_coordinates=True, load_self_train_data=True, self_train_data_dir=str(tmp_path),
                                 device=device, mode="train", volume_augmentations=_IdentityVolume(), color_drop=0.3)
    np.random.seed(seed)
    pyrandom.seed(seed)
    item = reader[0]
    rng_after = np.array([np.random.random(), pyrandom.random()])
    assert np.array_equal(rng_after, z[f"train{seed}/rng_after"])               # same draws, same order
    coords, feats = item[0].cpu().numpy(), item[1].cpu().numpy()
    ref_c, ref_f = z[f"train{seed}/coordinates"], z[f"train{seed}/features"]
    assert coords.dtype == ref_c.dtype and coords.shape == ref_c.shape
    # centring (numpy's ordered f32 column sums restated), shift, flips: bit for bit; the elastic displacement is
    # scipy-exact up to one rounding on < 0.1 % of the coordinates (tests/test_augment.py)
    np.testing.assert_allclose(coords, ref_c, rtol=0, atol=5e-7)
    assert float(np.mean(coords == ref_c)) > 0.995
    np.testing.assert_allclose(feats[:, :6], ref_f[:, :6], rtol=1e-6, atol=1e-6)        # colours | normals
    np.testing.assert_allclose(feats[:, 6:], ref_f[:, 6:], rtol=0, atol=5e-7)           # raw (augmented) coordinates
    assert np.array_equal(item[2], z[f"train{seed}/freemasks"])
    assert np.array_equal(item[6], z["item/raw_coordinates"])


@pytest.mark.gpu
def test_volume_and_colour_pipelines_match_their_definitions(device):
    """Scale3d + three RotateAroundAxis3d (conf/augmentation/volumentations_aug.yaml) as one affine pass, and
    RandomBrightnessContrast + RGBShift + Normalize (albumentations_aug.yaml) as one table pass, against step-by-step
    numpy restatements of the published transform definitions with the same drawn parameters."""
    from unscene3d_amd.datasets import augment as A

    rng = np.random.default_rng(0)
    pts = rng.uniform(-3, 3, (5000, 3)).astype(np.float32)
    nrm = rng.normal(0, 1, (5000, 3)).astype(np.float32)
    vol = A.VolumeAugmentations()
    scale, R = vol.draw()
    dp, dn = torch.from_numpy(pts).to(device), torch.from_numpy(nrm).to(device)
    vol(points=dp, normals=dn, features=None, labels=None, params=(scale, R))
    exp_p = (pts.astype(np.float64) * scale) @ R.T
    exp_n = nrm.astype(np.float64) @ R.T
    np.testing.assert_allclose(dp.cpu().numpy(), exp_p, rtol=0, atol=1e-6)
    np.testing.assert_allclose(dn.cpu().numpy(), exp_n, rtol=0, atol=1e-6)
    assert abs(np.linalg.det(R) - 1) < 1e-12 and np.allclose(R @ R.T, np.eye(3))

    col = rng.uniform(0, 255.99, (5000, 3)).astype(np.float32)
    ca = A.ColorAugmentations()
    alpha, beta, shifts = ca.draw()
    tables = ca.tables((alpha, beta, shifts))
    mean255 = np.array([0.478, 0.430, 0.375], np.float32) * 255
    inv = np.reciprocal(np.array([0.283, 0.276, 0.270], np.float32) * 255)
    got = A.color_tables_to_features(torch.from_numpy(col).to(device), tables, mean255, inv).cpu().numpy()
    img = col.astype(np.uint8)                                                   # the reference's pseudo image
    bc = np.clip(img.astype(np.float32) * np.float32(alpha) + np.float32(beta * 255.0), 0, 255).astype(np.uint8)
    sh = np.stack([np.clip(bc[:, c].astype(np.float32) + np.float32(shifts[c]), 0, 255).astype(np.uint8) for c in range(3)], 1)
    exp = (sh.astype(np.float32) - mean255) * inv
    np.testing.assert_allclose(got, exp, rtol=0, atol=1e-6)
