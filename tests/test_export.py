"""Export post-processing against golden vectors recorded from the reference's own
`InstanceSegmentation.eval_instance_step` (tests/golden/export.npz, generator: make_golden.py export).

CPU: the host logic (top-k, score sort, greedy overlap filter, label offset, boxes) with the three device row
operators replaced by torch CPU stand-ins.  GPU: the product path (HIP gather / segment mean / eps-components)."""
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "export.npz")
CASES = ("freemask", "dbscan", "plain")


def _load(z, name):
    scenes = []
    for i in range(2):
        g = lambda k: z[f"{name}/{i}/{k}"]
        n_full, k = (int(v) for v in g("out_n"))
        scenes.append(dict(
            raw_coords=g("raw_coords"), point2segment=g("point2segment"), pred_masks=g("pred_masks"),
            pred_logits=g("pred_logits"), inverse_map=g("inverse_map"), point2segment_full=g("point2segment_full"),
            full_res_coords=g("full_res_coords"),
            out_masks=np.unpackbits(g("out_masks"), axis=0)[:n_full].astype(bool).reshape(n_full, k),
            out_scores=g("out_scores"), out_classes=g("out_classes"), out_boxes=g("out_boxes")))
    general = NS(use_dbscan=bool(z[f"{name}/use_dbscan"]), dbscan_eps=0.95, topk_per_image=int(z[f"{name}/topk_per_image"]),
                 filter_out_instances=bool(z[f"{name}/filter_out_instances"]), scores_threshold=0.1, iou_threshold=0.66)
    return scenes, general


def _run(scenes, general, device):
    from unscene3d_amd.trainer import postprocess as PP

    dev = torch.device(device)
    output = {"aux_outputs": [],
              "pred_logits": torch.from_numpy(np.stack([s["pred_logits"] for s in scenes])).to(dev),
              "pred_masks": [torch.from_numpy(s["pred_masks"]).to(dev) for s in scenes]}
    low = [{"point2segment": torch.from_numpy(s["point2segment"]).to(dev)} for s in scenes]
    full = [{"point2segment": torch.from_numpy(s["point2segment_full"]).to(dev)} for s in scenes]
    return PP.export_instances(output, low, full, [s["inverse_map"] for s in scenes],
                               np.concatenate([s["raw_coords"] for s in scenes]), general, num_classes=3,
                               label_offset=2, full_res_coords=[s["full_res_coords"] for s in scenes])


def _check(results, scenes):
    for res, s in zip(results, scenes):
        masks = res["pred_masks"].cpu().numpy()
        assert masks.shape == s["out_masks"].shape
        assert np.array_equal(masks, s["out_masks"])
        np.testing.assert_allclose(res["pred_scores"], s["out_scores"], rtol=2e-5)
        assert np.array_equal(np.asarray(res["pred_classes"]), s["out_classes"])
        assert res["pred_boxes"].shape == s["out_boxes"].shape
        np.testing.assert_allclose(res["pred_boxes"], s["out_boxes"], rtol=1e-4, atol=1e-5)


def _cpu_row_ops(monkeypatch):
    """torch CPU stand-ins for the three HIP row operators (test scaffolding, not a product path)."""
    from sklearn.cluster import DBSCAN

    from unscene3d_amd import ops

    def gather_rows(src, idx):
        return src[idx]

    def segment_csr(seg, S):
        return ops.SegmentCSR(seg, None, None, S)

    def segment_mean(src, csr):
        out = torch.zeros((csr.S, src.shape[1])).index_add_(0, csr.seg, src)
        cnt = torch.zeros(csr.S).index_add_(0, csr.seg, torch.ones(src.shape[0]))
        return out / cnt.clamp(min=1)[:, None]

    def cc_eps(xyz, eps):
        return torch.from_numpy(DBSCAN(eps=eps, min_samples=1).fit(xyz.double().numpy()).labels_)

    for name, fn in (("gather_rows", gather_rows), ("segment_csr", segment_csr), ("segment_mean", segment_mean),
                     ("cc_eps", cc_eps)):
        monkeypatch.setattr(ops, name, fn)


@pytest.mark.parametrize("name", CASES)
def test_export_host_logic_matches_reference(name, monkeypatch):
    _cpu_row_ops(monkeypatch)
    scenes, general = _load(np.load(GOLD), name)
    results = _run(scenes, general, "cpu")
    assert (results[0]["pred_masks"].shape[1] > 0)
    _check(results, scenes)


@pytest.mark.parametrize("name", CASES)
def test_export_oracle_matches_reference_golden(name):
    """oracle/export_ref.py (the checker of the eval-mode tests) pinned by the reference's own eval_instance_step."""
    from oracle.export_ref import export_instances_ref

    scenes, general = _load(np.load(GOLD), name)
    t = torch.from_numpy
    res = export_instances_ref(t(np.stack([s["pred_logits"] for s in scenes])), [t(s["pred_masks"]) for s in scenes],
                               [t(s["point2segment"]).long() for s in scenes], [t(s["inverse_map"]).long() for s in scenes],
                               [t(s["point2segment_full"]).long() for s in scenes],
                               np.concatenate([s["raw_coords"] for s in scenes]), general, num_classes=3, label_offset=2)
    for r, s in zip(res, scenes):
        assert np.array_equal(r["pred_masks"], s["out_masks"])
        np.testing.assert_allclose(r["pred_scores"], s["out_scores"], rtol=2e-5)
        assert np.array_equal(r["pred_classes"], s["out_classes"])


@pytest.mark.parametrize("use_dbscan", [False, True])
def test_export_with_no_instances(use_dbscan, monkeypatch):
    """Every query mask empty: an empty result, not an exception (the reference's torch.stack raises)."""
    from unscene3d_amd.trainer import postprocess as PP

    _cpu_row_ops(monkeypatch)
    rng = np.random.default_rng(0)
    S, Q, N, NF = 20, 10, 200, 300
    general = NS(use_dbscan=use_dbscan, dbscan_eps=0.95, topk_per_image=100, filter_out_instances=True,
                 scores_threshold=0.1, iou_threshold=0.66)
    output = {"aux_outputs": [], "pred_logits": torch.randn(1, Q, 3), "pred_masks": [torch.full((S, Q), -5.0)]}
    low = [{"point2segment": torch.from_numpy(rng.integers(0, S, N))}]
    full = [{"point2segment": torch.from_numpy(rng.integers(0, S, NF))}]
    res = PP.export_instances(output, low, full, [rng.integers(0, N, NF)], rng.random((N, 3)), general, num_classes=3,
                              full_res_coords=[rng.random((NF, 3)).astype(np.float32)])[0]
    assert res["pred_masks"].shape == (NF, 0) and res["pred_scores"].shape == (0,) and res["pred_boxes"].shape == (0, 8)


def test_save_for_freemask_format(tmp_path):
    from unscene3d_amd.trainer import postprocess as PP

    coords = np.random.default_rng(0).random((50, 3)).astype(np.float32)
    masks = torch.rand(50, 4) > 0.5
    PP.save_for_freemask(str(tmp_path), "scene0000_00", coords, masks)
    m = np.load(tmp_path / "freemasks" / "scene0000_00_masks.npy")
    c = np.load(tmp_path / "freemasks" / "scene0000_00_cloud.npy")
    assert m.dtype == bool and np.array_equal(m, masks.numpy()) and np.array_equal(c, coords)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_export_device_path_matches_reference(device, name):
    scenes, general = _load(np.load(GOLD), name)
    _check(_run(scenes, general, device), scenes)


@pytest.mark.gpu
def test_export_runs_on_the_collate_output(device):
    """export_instances on exactly what FreeMaskVoxelizeCollate hands over (point2segment is a column of the
    [N, K+2] table there; the row-gather kernels need it dense): model-shaped random predictions, then every full-res
    mask must be constant per full-res segment and the masks of a single-query oracle must come back unchanged."""
    from unscene3d_amd.datasets.synthetic import SyntheticFreeMaskDataset
    from unscene3d_amd.datasets.utils import FreeMaskVoxelizeCollate
    from unscene3d_amd.trainer import postprocess as PP

    ds = SyntheticFreeMaskDataset(n_scenes=2, target_voxels=6000, seed=4100)
    collate = FreeMaskVoxelizeCollate(ignore_label=255, voxel_size=0.02, mode="validation", device=str(device))
    data, target, _ = collate([ds[0], ds[1]])
    Q = 12
    g = torch.Generator().manual_seed(0)
    masks, logits = [], []
    for t in target:
        assert t["point2segment"].is_contiguous()
        S = int(t["point2segment"].max()) + 1
        m = torch.randn(S, Q, generator=g)
        seg_mask = t["segment_mask"].float().cpu()                    # plant the targets as confident query masks
        k = min(Q, seg_mask.shape[0])
        m[:, :k] = (seg_mask[:k].T * 2 - 1) * 6
        masks.append(m.to(device))
        logits.append(torch.randn(Q, 3, generator=g))
    output = {"aux_outputs": [], "pred_logits": torch.stack(logits).to(device), "pred_masks": masks}
    general = NS(use_dbscan=False, dbscan_eps=0.95, topk_per_image=-1, filter_out_instances=False,
                 scores_threshold=0.1, iou_threshold=0.66)
    for eval_on_segments in (False, True):
        res = PP.export_instances(output, target, data.target_full, data.inverse_maps, None, general, num_classes=3,
                                  full_res_coords=data.full_res_coords, eval_on_segments=eval_on_segments)
        for bid, r in enumerate(res):
            full = r["pred_masks"]
            inv = data.inverse_maps[bid]
            assert full.shape[0] == inv.shape[0] and full.dtype == torch.bool and full.shape[1] == Q
            if eval_on_segments:
                continue
            # every exported column is one of the Q query masks lifted segment -> voxel -> full resolution
            lifted = (masks[bid] > 0)[target[bid]["point2segment"]][inv].T.cpu().numpy()      # [Q, N_full]
            have = {c.tobytes() for c in np.packbits(lifted, axis=1)}
            for c in np.packbits(full.T.cpu().numpy(), axis=1):
                assert c.tobytes() in have
