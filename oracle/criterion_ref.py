"""oracle/criterion_ref.py — CPU (torch) restatement of the set criterion of the self-training step:
`HungarianMatcher.memory_efficient_forward` (reference models/matcher.py:98-168, cost functions :12-63) and
`SetCriterion.forward / loss_labels / loss_masks` with `dice_loss` / `sigmoid_ce_loss` (models/criterion.py:22-73,
:138-216, :239-276).  TEST INFRASTRUCTURE ONLY: nothing under unscene3d_amd/ imports it, and it imports nothing
from unscene3d_amd/ — the "oracle step" of tests/test_gpu_step_parity.py no longer depends on product code.

Pinned by tests/golden/criterion.npz (the reference's own matcher + criterion imported and run in the build
container, tests/golden/make_golden.py): assignments exact, 12 losses 1e-5, gradients 1e-4
(tests/test_criterion_oracle.py).

Scope = the shipped self-training configuration (conf/loss/set_criterion.yaml, conf/matcher/hungarian_matcher.yaml):
num_points = -1 (every point), class_weights = -1, no DropLoss, cost_noise_robust = 0 (the tri-plane term is a
constant 0 entry per level, criterion.py:177).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment


def batch_dice_loss(inputs, targets):
    """matcher.py:12-27."""
    inputs = inputs.sigmoid().flatten(1)
    numerator = 2 * torch.einsum("nc,mc->nm", inputs, targets)
    denominator = inputs.sum(-1)[:, None] + targets.sum(-1)[None, :]
    return 1 - (numerator + 1) / (denominator + 1)


def batch_sigmoid_ce_loss(inputs, targets):
    """matcher.py:35-58."""
    hw = inputs.shape[1]
    pos = F.binary_cross_entropy_with_logits(inputs, torch.ones_like(inputs), reduction="none")
    neg = F.binary_cross_entropy_with_logits(inputs, torch.zeros_like(inputs), reduction="none")
    loss = torch.einsum("nc,mc->nm", pos, targets) + torch.einsum("nc,mc->nm", neg, (1 - targets))
    return loss / hw


@torch.no_grad()
def hungarian_match(outputs, targets, mask_type, cost_class=2.0, cost_mask=5.0, cost_dice=2.0):
    """matcher.py:98-168 with num_points = -1 -> [(query idx i64, target idx i64)] per scene."""
    bs, num_queries = outputs["pred_logits"].shape[:2]
    indices = []
    for b in range(bs):
        out_prob = outputs["pred_logits"][b].softmax(-1)
        tgt_ids = targets[b]["labels"].clone()
        ignore = tgt_ids == 253
        tgt_ids[ignore] = 0
        c_class = -out_prob[:, tgt_ids]
        c_class[:, ignore] = -1.0
        out_mask = outputs["pred_masks"][b].T.float()
        tgt_mask = targets[b][mask_type].to(out_mask).float()
        c_mask = batch_sigmoid_ce_loss(out_mask, tgt_mask)
        c_dice = batch_dice_loss(out_mask, tgt_mask)
        C = cost_mask * c_mask + cost_class * c_class + cost_dice * c_dice
        i, j = linear_sum_assignment(C.reshape(num_queries, -1).cpu())      # raises ValueError on NaN / inf
        indices.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
    return indices


def loss_labels(outputs, targets, indices, num_classes, eos_coef):
    """criterion.py:138-154; num_classes = number of object classes (the ctor's num_classes - 1)."""
    src_logits = outputs["pred_logits"].float()
    batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
    src_idx = torch.cat([src for src, _ in indices])
    target_classes_o = torch.cat([t["labels"][J] for t, (_, J) in zip(targets, indices)])
    target_classes = torch.full(src_logits.shape[:2], num_classes, dtype=torch.int64)
    target_classes[(batch_idx, src_idx)] = target_classes_o
    empty_weight = torch.ones(num_classes + 1, dtype=src_logits.dtype)
    empty_weight[-1] = eos_coef
    return {"loss_ce": F.cross_entropy(src_logits.transpose(1, 2), target_classes, empty_weight, ignore_index=253)}


def loss_masks(outputs, targets, indices, mask_type):
    """criterion.py:156-216 (num_points = -1, weights = 1, noise-robust term off)."""
    l_mask, l_dice, l_noise = [], [], []
    for b, (map_id, target_id) in enumerate(indices):
        pred = outputs["pred_masks"][b][:, map_id].T
        tgt = targets[b][mask_type][target_id].float().to(pred.dtype)
        num_masks = tgt.shape[0]                                             # :189 overwrites the normaliser
        bce = F.binary_cross_entropy_with_logits(pred, tgt, reduction="none")
        l_mask.append(bce.mean(1).sum() / num_masks)                         # sigmoid_ce_loss :51-68
        p = pred.sigmoid().flatten(1)                                        # dice_loss :22-43
        numerator = 2 * (p * tgt).sum(-1)
        denominator = p.sum(-1) + tgt.sum(-1)
        l_dice.append((1 - (numerator + 1) / (denominator + 1)).sum() / num_masks)
        l_noise.append(torch.as_tensor(0.0, dtype=torch.float32))
    return {"loss_mask": torch.sum(torch.stack(l_mask)), "loss_dice": torch.sum(torch.stack(l_dice)),
            "loss_noise_robust": torch.sum(torch.stack(l_noise))}


def set_criterion(outputs, targets, mask_type, num_classes=3, eos_coef=0.1, cost_class=2.0, cost_mask=5.0,
                  cost_dice=2.0, forced_indices=None, info=None):
    """SetCriterion.forward (criterion.py:239-276) -> {loss name: scalar} for the last level and every aux level.
    num_classes: the ctor argument (object classes + 1).  forced_indices: [level][scene] (query idx, target idx), last
    level FIRST then the aux levels in order — imposes another run's assignments (gradient parity tests); info: dict
    that receives the oracle's own assignments in that same layout."""
    levels = [{k: v for k, v in outputs.items() if k != "aux_outputs"}] + list(outputs.get("aux_outputs", []))
    own = [hungarian_match(lv, targets, mask_type, cost_class, cost_mask, cost_dice) for lv in levels] \
        if (forced_indices is None or info is not None) else None
    if info is not None:
        info["indices"] = own
    use = forced_indices if forced_indices is not None else own
    losses = {}
    for li, lv in enumerate(levels):
        idx = [(s.long().cpu(), t.long().cpu()) for s, t in use[li]]
        d = {}
        d.update(loss_labels(lv, targets, idx, num_classes - 1, eos_coef))
        d.update(loss_masks(lv, targets, idx, mask_type))
        sfx = "" if li == 0 else f"_{li - 1}"
        losses.update({k + sfx: v for k, v in d.items()})
    return losses
