"""Scene-list driver of the pseudo-mask generator (SURVEY.md §8e "replicas only"): static sharding by index mod W and
K scenes side by side on one GPU."""
import numpy as np
import pytest
import torch


def test_scene_shard_partitions_the_list():
    from unscene3d_amd.pseudo_masks.driver import scene_shard

    n = 1201                                                    # scenes of the ScanNet train split
    shards = [scene_shard(n, r, 8) for r in range(8)]
    assert sorted(i for s in shards for i in s) == list(range(n))
    assert all(s == list(range(r, n, 8)) for r, s in enumerate(shards))
    assert max(map(len, shards)) - min(map(len, shards)) <= 1
    assert scene_shard(3, 5, 8) == [] and scene_shard(5, 0, 1) == [0, 1, 2, 3, 4]
    with pytest.raises(ValueError):
        scene_shard(10, 8, 8)


@pytest.mark.gpu
def test_concurrent_scenes_give_the_sequential_masks(device):
    """Five different scenes, three in flight (own host thread + HIP stream each) == one after the other, and the two
    ranks of a world of 2 cover the list between them."""
    from unscene3d_amd.pseudo_masks.driver import PseudoMaskDriver
    from unscene3d_amd.synthetic import make_segment_scene

    scenes = []
    for seed in range(5):
        feats, conn, _ = make_segment_scene(100 + seed, side=14 + seed, dims=(64, 32), n_objects=6 + seed)
        S = feats[0].shape[0]
        scenes.append({"features": tuple(torch.from_numpy(f).to(device) for f in feats),
                       "unique_segments": torch.arange(S), "seg_connectivity": torch.from_numpy(conn)})
    clone = lambda: [dict(s, features=tuple(f.clone() for f in s["features"])) for s in scenes]
    seq = PseudoMaskDriver(device=device, concurrent=1).run(clone())
    par = PseudoMaskDriver(device=device, concurrent=3).run(clone())
    assert sorted(seq) == sorted(par) == list(range(5))
    for i in range(5):
        assert seq[i].shape[0] >= 1 and np.array_equal(seq[i], par[i]), i
    r0 = PseudoMaskDriver(device=device, concurrent=2, rank=0, world=2).run(clone())
    r1 = PseudoMaskDriver(device=device, concurrent=2, rank=1, world=2).run(clone())
    assert sorted(r0) == [0, 2, 4] and sorted(r1) == [1, 3]
    for i, m in {**r0, **r1}.items():
        assert np.array_equal(m, seq[i])


@pytest.mark.gpu
def test_scene_list_tool_end_to_end(device, tmp_path, capsys):
    """tools/pseudo_masks_run.py (the reference's `main`, pseudo_masks/unscene3d_pseudo_main.py:532-667): scene files
    -> per-segment features -> masked NCut -> voxel level -> full-resolution lift -> `_cloud.npy` / `_masks.npy`;
    a second run skips what is already there; the masks respect the over-segmentation."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pseudo_masks_run", os.path.join(root, "tools", "pseudo_masks_run.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    out = str(tmp_path / "pm")
    assert tool.main(["--synthetic", "3", "--out", out]) == 0
    first = capsys.readouterr().out
    assert "3 of 3 scenes" in first
    scenes = sorted(f for f in os.listdir(os.path.join(out, "scenes")) if f.endswith(".npz"))
    assert len(scenes) == 3
    for f in scenes:
        name = f[:-4]
        z = np.load(os.path.join(out, "scenes", f))
        cloud = np.load(os.path.join(out, f"{name}_cloud.npy"))
        masks = np.load(os.path.join(out, f"{name}_masks.npy"))
        assert cloud.dtype == np.float32 and cloud.shape == z["full_res_coords"].shape
        assert masks.dtype == bool and masks.shape[0] == cloud.shape[0] and 1 <= masks.shape[1] <= 20
        assert masks.any(0).all()                                   # no empty mask
        # a mask is a union of segments: every voxel of a segment carries the same bits
        vox = np.floor(cloud / 0.02).astype(np.int64)
        key = {tuple(c): s for c, s in zip(z["coords"][:, -3:].tolist(), z["segment_ids"].tolist())}
        seg_of_point = np.array([key.get(tuple(v), -1) for v in vox.tolist()])
        for s in np.unique(seg_of_point[seg_of_point >= 0])[:50]:
            rows = masks[seg_of_point == s]
            assert (rows == rows[0]).all()
    assert tool.main(["--synthetic", "3", "--out", out]) == 0
    assert "0 of 3 scenes" in capsys.readouterr().out               # everything was skipped


@pytest.mark.gpu
def test_long_run_of_trivial_scenes_does_not_recurse(device):
    """5 000 scenes that finish on their first step (the < 3-segment early exit of the NCut loop returns at once): the
    slot hand-over is a loop — the round-3 recursion would have exceeded Python's stack here."""
    import sys

    from unscene3d_amd.pseudo_masks.driver import PseudoMaskDriver

    def steps(scene):
        return scene * 2
        yield                                                   # noqa: a generator that ends before its first yield

    n = max(5000, 2 * sys.getrecursionlimit())
    out = PseudoMaskDriver(device=device, concurrent=4, scene_steps=steps).run(list(range(n)))
    assert len(out) == n and out[n - 1] == 2 * (n - 1)
