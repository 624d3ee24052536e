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
