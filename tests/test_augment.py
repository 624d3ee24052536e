"""Elastic distortion against golden vectors of the reference's own
`datasets.semseg.elastic_distortion` applied twice like freemask_semseg.py:356-361
(tests/golden/elastic.npz, generator: make_golden.py elastic; the noise grids numpy drew are part of the fixture)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "elastic.npz")
STEPS = ((0.2, 0.4), (0.8, 1.6))


def test_noise_grid_and_smoothing_follow_the_reference_arithmetic():
    import scipy.ndimage

    from unscene3d_amd.datasets.augment import noise_grid, smooth_noise

    z = np.load(GOLD)
    for name in ("f32", "f64"):
        pts = z[f"{name}/points"]
        dim, axes = noise_grid(pts[:, :3].min(0), pts[:, :3].max(0), 0.2)
        assert tuple(dim) == z[f"{name}/noise0"].shape[:3]            # the grid the reference drew for this cloud
        assert all(len(a) == d and a.dtype == np.float64 for a, d in zip(axes, dim))
    noise = z["f32/noise0"]
    ref = noise.copy()
    for _ in range(2):
        for shape in ((3, 1, 1, 1), (1, 3, 1, 1), (1, 1, 3, 1)):
            ref = scipy.ndimage.convolve(ref, np.ones(shape, np.float32) / 3, mode="constant", cval=0)
    assert np.array_equal(smooth_noise(torch.from_numpy(noise)).numpy(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["f32", "f64"])
def test_elastic_distortion_matches_reference(device, name):
    from unscene3d_amd.datasets.augment import elastic_distortion

    z = np.load(GOLD)
    pts = torch.from_numpy(z[f"{name}/points"]).to(device)
    for j, (granularity, magnitude) in enumerate(STEPS):
        out = elastic_distortion(pts, granularity, magnitude, noise=z[f"{name}/noise{j}"])
        assert out.data_ptr() == pts.data_ptr()                      # in place, like the reference
    got, ref = pts.cpu().numpy(), z[f"{name}/result"]
    assert got.dtype == ref.dtype
    assert np.array_equal(got[:, 3:], ref[:, 3:])                    # other columns untouched
    tol = 2e-6 if name == "f32" else 1e-12
    np.testing.assert_allclose(got[:, :3], ref[:, :3], rtol=0, atol=tol)
    assert (got[:, :3] == ref[:, :3]).mean() > 0.999                 # same operation order: (nearly) every bit


@pytest.mark.gpu
def test_elastic_distortion_consumes_numpys_stream_like_the_reference(device):
    from unscene3d_amd.datasets.augment import elastic_distortion

    z = np.load(GOLD)
    pts = torch.from_numpy(z["f32/points"]).to(device)
    np.random.seed(1234)                                             # the generator's seed: same draws, same result
    for granularity, magnitude in STEPS:
        elastic_distortion(pts, granularity, magnitude)
    np.testing.assert_allclose(pts.cpu().numpy()[:, :3], z["f32/result"][:, :3], rtol=0, atol=2e-6)
