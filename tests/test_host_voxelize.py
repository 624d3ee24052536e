"""CPU: the HIP-free voxelisation entry points (usc_voxel_floor_f64_host, usc_unique_coords_host) behind
ME.utils.sparse_quantize(device="cpu") — what the reference calls from FORKED DataLoader workers
(datasets/utils.py:403-414, conf/data/indoor.yaml:24 num_workers = 4) — against the oracle, and from such workers."""
import numpy as np
import pytest
import torch

from oracle import sparse_ref as R
from unscene3d_amd import MinkowskiEngine as ME


def _scene(seed, n):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-4.0, 4.0, (n, 3))
    xyz[: n // 3] = xyz[n // 3: 2 * (n // 3)] + rng.normal(0, 0.004, (n // 3, 3))     # many points per voxel
    return xyz


@pytest.mark.parametrize("n", [0, 1, 257, 20000, 150000])
def test_host_sparse_quantize_equals_the_oracle(n):
    xyz = _scene(n, n)
    ec = R.voxel_floor(xyz, 0.02)
    eu, einv = R.sparse_quantize(ec)
    feats = np.arange(n, dtype=np.float32)[:, None].repeat(2, 1)
    c, f, u, inv = ME.utils.sparse_quantize(xyz, features=feats, quantization_size=0.02, return_index=True,
                                            return_inverse=True, device="cpu")
    assert not c.is_cuda and c.dtype == torch.int32
    assert np.array_equal(u.numpy(), eu) and np.array_equal(inv.numpy(), einv)
    assert np.array_equal(c.numpy(), ec[eu]) and np.array_equal(f.numpy(), feats[eu])
    # already-floored integer coordinates, with and without the batch column (reference datasets/utils.py:403 floors first)
    u2, inv2 = ME.utils.sparse_quantize(ec, return_index=True, return_inverse=True, return_maps_only=True, device="cpu")
    assert np.array_equal(u2.numpy(), eu) and np.array_equal(inv2.numpy(), einv)
    if n:
        b = np.sort(np.random.default_rng(1).integers(0, 3, (n, 1)), 0).astype(np.int32)
        c4 = np.concatenate([b, ec], 1)
        e4u, e4inv = R.sparse_quantize(c4)
        u4, inv4 = ME.utils.sparse_quantize(c4, return_index=True, return_inverse=True, return_maps_only=True, device="cpu")
        assert np.array_equal(u4.numpy(), e4u) and np.array_equal(inv4.numpy(), e4inv)


def test_out_of_range_coordinate_is_an_error_not_a_crash():
    c = np.array([[0, 0, 0], [1 << 20, 0, 0]], dtype=np.int32)
    with pytest.raises(RuntimeError, match="packable range"):
        ME.utils.sparse_quantize(c, return_index=True, device="cpu")


class _Scenes(torch.utils.data.Dataset):
    def __len__(self):
        return 6

    def __getitem__(self, i):
        return _scene(100 + i, 5000 + 100 * i)


def _collate(batch):
    # the reference's collate step: floor + unique per scene, inside the worker process
    out = []
    for xyz in batch:
        c, u, inv = ME.utils.sparse_quantize(xyz, quantization_size=0.02, return_index=True, return_inverse=True,
                                             device="cpu")
        out.append((c, u, inv))
    return out


def test_forked_dataloader_workers_can_voxelise():
    """num_workers > 0 forks (Linux default): the workers call into libusc3d_hip.so without touching HIP."""
    loader = torch.utils.data.DataLoader(_Scenes(), batch_size=2, num_workers=2, collate_fn=_collate,
                                         multiprocessing_context="fork")
    seen = 0
    for bi, batch in enumerate(loader):
        for j, (c, u, inv) in enumerate(batch):
            i = 2 * bi + j
            ec = R.voxel_floor(_scene(100 + i, 5000 + 100 * i), 0.02)
            eu, einv = R.sparse_quantize(ec)
            assert np.array_equal(u.numpy(), eu) and np.array_equal(inv.numpy(), einv) and np.array_equal(c.numpy(), ec[eu])
            seen += 1
    assert seen == 6
