"""Evaluation-time mask splitting (reference trainer/trainer.py:507-539, `general.use_dbscan`): every query
mask is split into the connected components of its eps-ball graph.  The reference calls
sklearn DBSCAN(eps=0.95, min_samples=1) once per query; here the components come from the device
(`ops.cc_eps`, identical labels)."""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import ops


def dbscan_split(masks: torch.Tensor, logits: torch.Tensor, coords: torch.Tensor, eps: float = 0.95):
    """masks f32[N,Q] (per-point logits), logits [Q,C], coords f32[N,3] (device)
    -> (new_masks [N,Q'], new_logits [Q',C]) with one column per (query, cluster)."""
    new_masks, new_logits = [], []
    for q in range(masks.shape[1]):
        on = masks[:, q] > 0
        if int(on.sum()) == 0:
            continue
        labels = ops.cc_eps(coords[on].float().contiguous(), eps)
        full = torch.zeros(masks.shape[0], dtype=torch.int64, device=masks.device)
        full[on] = labels + 1
        for cid in range(int(labels.max()) + 1):
            new_masks.append(masks[:, q] * (full == cid + 1))
            new_logits.append(logits[q])
    if not new_masks:   # every query mask empty (the reference's torch.stack raises here): no instances
        return masks.new_zeros((masks.shape[0], 0)), logits.new_zeros((0, logits.shape[1]))
    return torch.stack(new_masks).T, torch.stack(new_logits)


# ------------------------------------------------------------------------------------------------------------------
# Export of predicted instances between self-training rounds (reference
# trainer/trainer.py:445-477 get_full_res_mask / get_mask_and_scores, :479-760 eval_instance_step).  Everything that
# touches [points, queries] arrays stays on the device; the [queries]-sized selections (top-k, score sort, the greedy
# overlap filter) run on the host exactly like the reference's CPU code, so ties break the same way.
def get_mask_and_scores(mask_cls, mask_pred, num_queries=100, num_classes=18, topk_per_image=-1):
    """mask_cls [Q,C] class probabilities, mask_pred [N,Q] mask logits (device)
    -> (score [k], result_pred_mask f32[N,k], classes i64[k], heatmap f32[N,k]); trainer.py:456-477.
    k is clamped to the number of (query, class) candidates; the reference's topk raises when asked for more."""
    dev = mask_pred.device
    k = topk_per_image if topk_per_image != -1 else num_queries
    k = min(k, mask_cls.shape[0] * mask_cls.shape[1])   # fewer candidates than asked for (e.g. after a DBSCAN split)
    labels = torch.arange(num_classes).unsqueeze(0).repeat(num_queries, 1).flatten(0, 1)
    scores_per_query, topk_indices = mask_cls.detach().cpu().flatten(0, 1).topk(k, sorted=True)
    classes = labels[topk_indices]
    cols = (topk_indices // num_classes).to(dev)
    mask_pred = mask_pred.float().index_select(1, cols)
    result_pred_mask = (mask_pred > 0).float()
    heatmap = mask_pred.sigmoid()
    mask_scores = (heatmap * result_pred_mask).sum(0) / (result_pred_mask.sum(0) + 1e-6)
    score = scores_per_query.to(dev) * mask_scores
    return score, result_pred_mask, classes, heatmap


def get_full_res_mask(mask, inverse_map, segments_full=None, is_heatmap=False):
    """mask f32[N_low,k] -> f32[N_full,k]: rows gathered through the voxelisation's inverse map; binary masks are
    then averaged per full-resolution segment, thresholded at 0.5 and broadcast back (trainer.py:445-453).
    segments_full: `ops.SegmentCSR` of point2segment_full (build once per scene), or None (eval_on_segments off)."""
    full = ops.gather_rows(mask.contiguous(), inverse_map)
    if segments_full is not None and not is_heatmap:
        seg_mean = ops.segment_mean(full, segments_full)
        full = ops.gather_rows((seg_mean > 0.5).float(), segments_full.seg)
    return full


def filter_instances(sorted_masks, sorted_scores, scores_threshold, iou_threshold):
    """Greedy overlap filter of trainer.py:586-607 on score-sorted binary masks f32[N,k] -> kept column indices.
    The k x k overlap counts are exact in fp32 (integers < 2^24) and the normalisation repeats numpy's fp32 ops."""
    if sorted_masks.shape[1] == 0:
        return []
    overlap = sorted_masks.T @ sorted_masks
    norm = overlap / (overlap.max(dim=0).values + 10e-8)
    over = (norm > iou_threshold).cpu().numpy()
    empty = (overlap.diagonal() == 0).cpu().numpy()
    keep = []
    for i in range(over.shape[0]):
        if sorted_scores[i] < scores_threshold or empty[i]:
            continue
        ids = over[i].nonzero()[0]
        if len(ids) == 0 or i == ids.min():
            keep.append(i)
    return keep


def export_instances(output, target_low_res, target_full_res, inverse_maps, raw_coords, general, num_classes,
                     decoder_id=-1, train_on_segments=True, eval_on_segments=True, label_offset=0,
                     full_res_coords=None):
    """Device version of the prediction half of `eval_instance_step` (trainer.py:479-651, :668-683).

    output: model output dict (`pred_logits` [B,Q,C+1], `pred_masks` list of [S_b or N_b, Q], `aux_outputs`);
    target_*[b]['point2segment']; inverse_maps[b] i64[N_full_b]; raw_coords f32[sum N_b, 3] (DBSCAN only);
    general: config node with use_dbscan, dbscan_eps, topk_per_image, filter_out_instances, scores_threshold,
    iou_threshold.  -> list of dicts {pred_masks bool[N_full,K], pred_scores f32[K], pred_classes i64[K]
    (+ label_offset), pred_boxes f64[K',8] = class, centre, extent, score of the non-empty masks}."""
    preds = list(output["aux_outputs"]) + [{"pred_logits": output["pred_logits"], "pred_masks": output["pred_masks"]}]
    pred = preds[decoder_id]
    logits = torch.softmax(pred["pred_logits"], dim=-1)[..., :-1]
    results, offset = [], 0
    for bid in range(len(pred["pred_masks"])):
        dev = pred["pred_masks"][bid].device
        masks = pred["pred_masks"][bid].detach().float()
        if train_on_segments:
            p2s = torch.as_tensor(target_low_res[bid]["point2segment"]).to(dev, torch.int64).contiguous()
            masks = ops.gather_rows(masks.contiguous(), p2s)
        if general.use_dbscan:
            n = masks.shape[0]
            coords = torch.as_tensor(raw_coords[offset:offset + n], dtype=torch.float32, device=dev)
            offset += n
            new_masks, new_logits = dbscan_split(masks, logits[bid], coords, eps=general.dbscan_eps)
            scores, masks, classes, heatmap = get_mask_and_scores(new_logits, new_masks, new_logits.shape[0],
                                                                  num_classes - 1, general.topk_per_image)
        else:
            scores, masks, classes, heatmap = get_mask_and_scores(logits[bid], masks, logits.shape[1],
                                                                  num_classes - 1, general.topk_per_image)
        inv = torch.as_tensor(inverse_maps[bid], dtype=torch.int64, device=dev).contiguous()
        seg_full = torch.as_tensor(target_full_res[bid]["point2segment"], dtype=torch.int64, device=dev).contiguous()
        csr = ops.segment_csr(seg_full, int(seg_full.max()) + 1) if eval_on_segments else None
        masks = get_full_res_mask(masks, inv, csr)
        order = scores.cpu().sort(descending=True)
        idx = order.indices
        sorted_scores = order.values.numpy()
        sorted_classes = classes[idx]
        sorted_masks = masks.index_select(1, idx.to(dev))
        if general.filter_out_instances:
            keep = filter_instances(sorted_masks, sorted_scores, general.scores_threshold, general.iou_threshold)
            sorted_masks = sorted_masks.index_select(1, torch.as_tensor(keep, dtype=torch.int64, device=dev))
            sorted_scores, sorted_classes = sorted_scores[keep], sorted_classes[keep]
        res = {"pred_masks": sorted_masks > 0, "pred_scores": sorted_scores, "pred_classes": sorted_classes + label_offset}
        if full_res_coords is not None:
            res["pred_boxes"] = mask_boxes(res["pred_masks"], torch.as_tensor(full_res_coords[bid], device=dev),
                                           res["pred_classes"], sorted_scores)
        results.append(res)
    return results


def mask_boxes(masks, coords, classes, scores):
    """(class, centre xyz, extent xyz, score) of every non-empty mask column (trainer.py:668-683) -> f64[K',8]."""
    m = masks.float()
    cnt = m.sum(0)
    centre = (m.T @ coords.float()) / cnt.clamp(min=1)[:, None]
    big = torch.finfo(torch.float32).max
    ext = []
    for a in range(3):                                      # one [N,K] temporary per axis, not [N,K,3]
        ca = coords[:, a:a + 1].float()
        ext.append(torch.where(masks, ca, -big).amax(0) - torch.where(masks, ca, big).amin(0))
    rows = torch.cat([centre, torch.stack(ext, 1)], 1).double().cpu().numpy()
    ok = (cnt > 0).cpu().numpy()
    out = np.concatenate([np.asarray(classes, np.float64)[:, None], rows, np.asarray(scores, np.float64)[:, None]], 1)
    return out[ok]


def save_for_freemask(save_dir, file_name, full_res_coords, pred_masks):
    """`{save_dir}/freemasks/{name}_cloud.npy` (coordinates) and `{name}_masks.npy` (bool [N_full,K]) — the files
    the next self-training round's preprocessing reads (trainer.py:743-760)."""
    d = os.path.join(save_dir, "freemasks")
    os.makedirs(d, exist_ok=True)
    np.save(os.path.join(d, f"{file_name}_cloud.npy"), np.asarray(full_res_coords))
    masks = pred_masks.cpu().numpy() if torch.is_tensor(pred_masks) else np.asarray(pred_masks)
    np.save(os.path.join(d, f"{file_name}_masks.npy"), masks.astype(bool))
