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
