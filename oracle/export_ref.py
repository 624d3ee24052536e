"""oracle/export_ref.py — CPU (torch + numpy) restatement of the prediction half of
`InstanceSegmentation.eval_instance_step` (reference trainer/trainer.py:479-651) with its helpers
`get_mask_and_scores` (:456-477) and `get_full_res_mask` (:445-453).  TEST INFRASTRUCTURE ONLY: nothing under
unscene3d_amd/ imports it.

Pinned by tests/golden/export.npz — outputs of the reference's own `eval_instance_step` imported and run in the build
container (tests/golden/make_golden.py export): kept masks and classes bit-exact, scores 2e-5
(tests/test_export.py::test_export_oracle_matches_reference_golden).  It is the checker for the model outputs the
DEVICE produces in eval mode (tests/test_gpu_eval_parity.py): the same function applied to the oracle's forward gives
the masks the reference would export for the next self-training round.
"""
from __future__ import annotations

import numpy as np
import torch


def scatter_mean(src: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """torch_scatter.scatter_mean(src, index, dim=0): sum / max(count, 1)."""
    S = int(index.max()) + 1
    out = torch.zeros((S, src.shape[1]), dtype=src.dtype).index_add_(0, index, src)
    cnt = torch.zeros(S, dtype=src.dtype).index_add_(0, index, torch.ones(src.shape[0], dtype=src.dtype))
    return out / cnt.clamp(min=1)[:, None]


def get_mask_and_scores(mask_cls, mask_pred, num_queries, num_classes, topk_per_image):
    """trainer.py:456-477."""
    labels = torch.arange(num_classes).unsqueeze(0).repeat(num_queries, 1).flatten(0, 1)
    k = topk_per_image if topk_per_image != -1 else num_queries
    scores_per_query, topk_indices = mask_cls.flatten(0, 1).topk(k, sorted=True)
    labels_per_query = labels[topk_indices]
    topk_indices = topk_indices // num_classes
    mask_pred = mask_pred[:, topk_indices]
    result_pred_mask = (mask_pred > 0).float()
    heatmap = mask_pred.float().sigmoid()
    mask_scores = (heatmap * result_pred_mask).sum(0) / (result_pred_mask.sum(0) + 1e-6)
    return scores_per_query * mask_scores, result_pred_mask, labels_per_query, heatmap


def get_full_res_mask(mask, inverse_map, point2segment_full, eval_on_segments=True, is_heatmap=False):
    """trainer.py:445-453."""
    mask = mask[inverse_map]
    if eval_on_segments and not is_heatmap:
        mask = scatter_mean(mask, point2segment_full)
        mask = (mask > 0.5).float()
        mask = mask[point2segment_full]
    return mask


def export_instances_ref(pred_logits, pred_masks, point2segment, inverse_maps, point2segment_full, raw_coords, general,
                         num_classes, label_offset=0, train_on_segments=True, eval_on_segments=True):
    """pred_logits f32[B,Q,C+1], pred_masks list of f32[S_b,Q] (CPU tensors of the decoder level to export)
    -> list of dict(pred_masks bool[N_full,K], pred_scores f32[K], pred_classes i64[K]); trainer.py:490-607, :629."""
    logits = torch.softmax(pred_logits, dim=-1)[..., :-1]
    results, offset = [], 0
    for bid in range(len(pred_masks)):
        masks = pred_masks[bid].detach().float()
        if train_on_segments:
            masks = masks[point2segment[bid]]
        if general.use_dbscan:
            from sklearn.cluster import DBSCAN
            n = masks.shape[0]
            coords = np.asarray(raw_coords[offset:offset + n])
            offset += n
            new_masks, new_logits = [], []
            for q in range(masks.shape[1]):
                on = masks[:, q] > 0
                if int(on.sum()) > 0:
                    clusters = DBSCAN(eps=general.dbscan_eps, min_samples=1).fit(coords[on.numpy()]).labels_
                    new_mask = torch.zeros(on.shape, dtype=torch.int64)
                    new_mask[on] = torch.from_numpy(clusters) + 1
                    for cid in np.unique(clusters):
                        if cid != -1:
                            new_masks.append(masks[:, q] * (new_mask == cid + 1))
                            new_logits.append(logits[bid, q])
            scores, masks, classes, _ = get_mask_and_scores(torch.stack(new_logits), torch.stack(new_masks).T,
                                                            len(new_logits), num_classes - 1, general.topk_per_image)
        else:
            scores, masks, classes, _ = get_mask_and_scores(logits[bid], masks, logits.shape[1], num_classes - 1,
                                                            general.topk_per_image)
        masks = get_full_res_mask(masks, inverse_maps[bid], point2segment_full[bid], eval_on_segments).numpy()
        order = scores.sort(descending=True)
        idx, vals = order.indices.numpy(), order.values.numpy()
        classes = classes[idx]
        sorted_masks = masks[:, idx]
        if general.filter_out_instances:
            keep = set()
            overlap = sorted_masks.T @ sorted_masks
            norm = overlap / (overlap.max(axis=0) + 10e-8)
            for i in range(norm.shape[0]):
                if not (vals[i] < general.scores_threshold) and not sorted_masks[:, i].sum() == 0.0:
                    ids = set(np.nonzero(norm[i, :] > general.iou_threshold)[0])
                    if len(ids) == 0 or i == min(ids):
                        keep.add(i)
            keep = sorted(keep)
            sorted_masks, vals, classes = sorted_masks[:, keep], vals[keep], classes[keep]
        results.append({"pred_masks": sorted_masks > 0, "pred_scores": vals,
                        "pred_classes": classes.numpy() + label_offset})
    return results
