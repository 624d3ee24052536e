"""Feeders of the NCut core (unscene3d_amd/pseudo_masks/pipeline.py; reference
pseudo_masks/unscene3d_pseudo_main.py:287-348 `encode_scene_feats`, :649-667 the save-time lift): the two exact 1-NN
transfers against scipy's KD-tree (the reference's tool) and the per-frame 2D -> 3D running mean against the oracle."""
import numpy as np
import pytest
import torch
from scipy.spatial import KDTree

from oracle import project_ref as PR
from unscene3d_amd.synthetic import camera_views as cameras, room_voxels as room

pytestmark = pytest.mark.gpu
F = np.float32


def test_encode_scene_feats_3d_carries_coarse_features_to_the_input_voxels(device):
    """Features of level res_2 for every input voxel by exact 1-NN between voxel coordinates (:336-345).  Voxel centres
    of the stride-2 map sit on even integers, input voxels on all integers: nearest = own parent or a tie-free
    neighbour because ties (equidistant parents) are resolved identically only by chance — the test therefore checks
    the DISTANCE of every match against the KD-tree's and the features through the match."""
    from types import SimpleNamespace

    from unscene3d_amd import MinkowskiEngine as ME
    from unscene3d_amd.models.res16unet import Res16UNet34CMultiRes
    from unscene3d_amd.pseudo_masks.pipeline import encode_scene_feats_3d

    coords = room(5, batch=1)
    torch.manual_seed(0)
    cfg = SimpleNamespace(bn_momentum=0.02, conv1_kernel_size=3, dilations=[1, 1, 1, 1])
    model = Res16UNet34CMultiRes(3, 20, cfg).to(device).eval()
    feats = torch.rand(coords.shape[0], 3, device=device)
    x = ME.SparseTensor(features=feats, coordinates=torch.from_numpy(coords).to(device), device=device)
    out = encode_scene_feats_3d(model, x, resolution_scale=2)
    with torch.no_grad():
        _, fmaps = model(ME.SparseTensor(features=feats, coordinates=torch.from_numpy(coords).to(device), device=device))
    enc = fmaps["res_2"]
    lr = enc.C[:, 1:].float().cpu().numpy()
    hr = coords[:, 1:].astype(np.float32)
    d_ref, _ = KDTree(lr).query(hr, k=1)
    assert out.shape == (coords.shape[0], enc.F.shape[1])
    # every row of `out` is a row of the coarse feature map at the KD-tree's distance
    encF = enc.F.cpu().numpy()
    o = out.cpu().numpy()
    sample = np.random.default_rng(0).choice(coords.shape[0], 300, replace=False)
    for i in sample:
        cand = np.nonzero(np.abs(np.linalg.norm(lr - hr[i], axis=1) - d_ref[i]) < 1e-6)[0]
        assert any(np.array_equal(o[i], encF[j]) for j in cand), i


def test_masks_to_full_resolution_matches_kdtree(device):
    """Segment ids / masks of the voxels -> full-resolution points by 1-NN against voxel centres + 0.5 (:651-655)."""
    from unscene3d_amd.pseudo_masks.pipeline import masks_to_full_resolution

    rng = np.random.default_rng(3)
    coords = room(7, batch=1)
    n = coords.shape[0]
    seg = rng.integers(0, 40, n)
    masks = rng.random((n, 6)) < 0.3
    full = (coords[rng.integers(0, n, 5000), 1:] + rng.uniform(0.05, 0.95, (5000, 3))) * 0.02     # points inside voxels
    got_seg, got_masks = masks_to_full_resolution(torch.from_numpy(coords).to(device), full, 0.02, seg, masks)
    _, match = KDTree(coords[:, 1:].astype(np.float32) + 0.5).query((full / 0.02).astype(np.float32), k=1)
    assert np.array_equal(got_seg, seg[match]) and np.array_equal(got_masks, masks[match])


def test_encode_scene_feats_2d_running_mean_over_frames(device):
    """Image branch (:287-330): per frame a 2D model's key / query feature maps are cast onto the voxels and folded into
    the scene's running mean — against the oracle's first_hit / project_features / fuse_frame, bit for bit."""
    from unscene3d_amd import project_features_cuda as P
    from unscene3d_amd.pseudo_masks.pipeline import encode_scene_feats_2d

    W, H, C, n_frames = 20, 14, 48, 3
    DMIN, DMAX, INC = 0.1 / 0.02, 0.9 / 0.02, 0.01
    coords = room(21, batch=1)
    n = coords.shape[0]
    occ, shifts = PR.dense_occupancy(coords)
    raw_views = cameras(9, coords, n_frames)                                   # [1, n_frames, 4, 4]
    intr = np.tile(np.array([[W * 0.9, W * 0.9, (W - 1) / 2 + 0.25, (H - 1) / 2 - 0.4]], F), (1, 1))
    rng = np.random.default_rng(2)
    key = rng.standard_normal((n_frames, 1, 1, H, W, C)).astype(F)
    qry = rng.standard_normal((n_frames, 1, 1, H, W, C)).astype(F)
    images = torch.arange(n_frames, dtype=torch.float32, device=device).view(1, n_frames, 1, 1, 1)   # frame id as "image"

    def model(img):                                                             # the 2D backbone is the caller's
        i = int(img.flatten()[0].item())
        return torch.from_numpy(key[i]).to(device), torch.from_numpy(qry[i]).to(device)

    proj = P.Project2DFeaturesCUDA(width=W, height=H, voxel_size=0.02, depth_min=DMIN * 0.02, depth_max=DMAX * 0.02)
    sk, sq = encode_scene_feats_2d(model, images, torch.from_numpy(raw_views).to(device), torch.from_numpy(intr).to(device),
                                   torch.from_numpy(coords).to(device), proj, attention=True)
    ref_k, ref_q = np.zeros((n, C), F), np.zeros((n, C), F)
    for v in range(n_frames):
        h = PR.first_hit(occ, PR.shift_views(raw_views[:1, v:v + 1], shifts), intr, W, H, DMIN, DMAX, INC)
        pk, nk = PR.project_features(key[v], h, n)
        pq, nq = PR.project_features(qry[v], h, n)
        ref_k, ref_q = PR.fuse_frame(ref_k, pk, nk), PR.fuse_frame(ref_q, pq, nq)
    assert np.array_equal(sk.cpu().numpy(), ref_k) and np.array_equal(sq.cpu().numpy(), ref_q)
    only_key = encode_scene_feats_2d(model, images, torch.from_numpy(raw_views).to(device), torch.from_numpy(intr).to(device),
                                     torch.from_numpy(coords).to(device), proj, attention=False)
    assert np.array_equal(only_key.cpu().numpy(), ref_k)
