"""SetCriterion (reference models/criterion.py:94-292): Hungarian matching per prediction level,
weighted cross-entropy over C+1 classes (`eos_coef` on the no-object class, ignore_index 253),
sigmoid-BCE + dice on the matched masks, each divided by the number of target masks.

Differences from the reference are organisational only: all 13 levels' cost matrices are built on
the device first and copied to the host in ONE transfer (the reference syncs 13*B times), then
scipy solves them; the loss arithmetic and its reduction order follow the reference line by line."""
import torch
import torch.nn.functional as F
from torch import nn

from .misc import get_world_size, is_dist_avail_and_initialized


def dice_loss(inputs, targets, num_masks: float, weights):
    p = inputs.sigmoid().flatten(1)
    num = 2 * (p * targets).sum(-1)
    den = p.sum(-1) + targets.sum(-1)
    return (weights * (1 - (num + 1) / (den + 1))).sum() / num_masks


def sigmoid_ce_loss(inputs, targets, num_masks: float, weights):
    loss = weights.view(-1, 1) * F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    return loss.mean(1).sum() / num_masks


dice_loss_jit = dice_loss
sigmoid_ce_loss_jit = sigmoid_ce_loss


class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses, num_points, oversample_ratio,
                 importance_sample_ratio, class_weights, directions="xyz", use_droploss=False,
                 droploss_iou_thresh=0.1):
        super().__init__()
        self.num_classes = num_classes - 1
        self.class_weights, self.matcher, self.weight_dict = class_weights, matcher, weight_dict
        self.eos_coef, self.losses = eos_coef, list(losses)
        self.use_droploss, self.droploss_iou_thresh = use_droploss, droploss_iou_thresh
        empty_weight = torch.ones(self.num_classes + 1)
        empty_weight[-1] = self.eos_coef
        if self.class_weights != -1:
            assert len(self.class_weights) == self.num_classes, "CLASS WEIGHTS DO NOT MATCH"
            empty_weight[:-1] = torch.tensor(self.class_weights)
        self.register_buffer("empty_weight", empty_weight)
        self.num_points, self.oversample_ratio = num_points, oversample_ratio
        self.importance_sample_ratio = importance_sample_ratio
        self.directions = directions
        self.noise_robust_projection_loss = None    # built lazily: only when loss_noise_robust != 0

    # -- individual losses ------------------------------------------------------------------
    def loss_labels(self, outputs, targets, indices, num_masks, mask_type, coords=None):
        src_logits = outputs["pred_logits"].float()
        batch_idx, src_idx = self._get_src_permutation_idx(indices)
        matched = torch.cat([t["labels"][J.to(t["labels"].device)] for t, (_, J) in zip(targets, indices)])
        target_classes = torch.full(src_logits.shape[:2], self.num_classes, dtype=torch.int64,
                                    device=src_logits.device)
        target_classes[batch_idx.to(src_logits.device), src_idx.to(src_logits.device)] = matched.to(src_logits.device)
        loss_ce = F.cross_entropy(src_logits.transpose(1, 2), target_classes, self.empty_weight, ignore_index=253)
        return {"loss_ce": loss_ce}

    def loss_masks(self, outputs, targets, indices, num_masks, mask_type="masks", coords=None):
        l_mask, l_dice, l_noise = [], [], []
        for b, (map_id, target_id) in enumerate(indices):
            pred = outputs["pred_masks"][b]
            dev = pred.device
            m = pred[:, map_id.to(dev)].T                                   # [T, S]
            tgt = targets[b][mask_type][target_id.to(targets[b][mask_type].device)]
            if self.weight_dict["loss_noise_robust"] != 0:
                from .noise_robust_loss import ProjectionMaskLoss
                if self.noise_robust_projection_loss is None:
                    self.noise_robust_projection_loss = ProjectionMaskLoss(directions=self.directions)
                sampled = m[:, targets[b]["point2segment"]] if coords.shape[0] != m.shape[1] else m
                bmask = coords[:, 0] == b
                bl, all_shape = self.noise_robust_projection_loss(
                    sampled, targets[b]["masks"][target_id.to(targets[b]["masks"].device)].float(), coords[bmask])
                l_noise.append(bl / all_shape)
            else:
                l_noise.append(torch.as_tensor(0.0, dtype=torch.float32, device=dev))
            if self.num_points != -1:
                pidx = torch.randperm(tgt.shape[1], device=tgt.device)[:int(self.num_points * tgt.shape[1])]
                m, tgt = m[:, pidx], tgt[:, pidx]
            n_here = tgt.shape[0]
            if self.use_droploss:
                fg = m > 0.0
                iou = (fg * tgt).sum(dim=1) / (fg + tgt).sum(dim=1)
                weights = (iou >= self.droploss_iou_thresh).float()
            else:
                weights = torch.ones(m.shape[0], device=dev)
            tgt = tgt.float()
            l_mask.append(sigmoid_ce_loss(m, tgt, n_here, weights))
            l_dice.append(dice_loss(m, tgt, n_here, weights))
        return {"loss_mask": torch.sum(torch.stack(l_mask)), "loss_dice": torch.sum(torch.stack(l_dice)),
                "loss_noise_robust": torch.sum(torch.stack(l_noise))}

    @staticmethod
    def _get_src_permutation_idx(indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        return batch_idx, torch.cat([src for (src, _) in indices])

    @staticmethod
    def _get_tgt_permutation_idx(indices):
        batch_idx = torch.cat([torch.full_like(tgt, i) for i, (_, tgt) in enumerate(indices)])
        return batch_idx, torch.cat([tgt for (_, tgt) in indices])

    def get_loss(self, loss, outputs, targets, indices, num_masks, mask_type, coords=None):
        table = {"labels": self.loss_labels, "masks": self.loss_masks}
        assert loss in table, f"do you really want to compute {loss} loss?"
        return table[loss](outputs, targets, indices, num_masks, mask_type, coords)

    # -- matching of every level with one host transfer ------------------------------------
    def match_all_levels(self, levels, targets, mask_type):
        per_level = [self.matcher.cost_matrices(lv, targets, mask_type) for lv in levels]
        flat = [C for Cs in per_level for C in Cs]
        widths = [C.shape[1] for C in flat]
        host = torch.cat(flat, dim=1).cpu()                                   # the single D2H of the step
        pieces = torch.split(host, widths, dim=1)
        out, p = [], 0
        for Cs in per_level:
            out.append([self.matcher.solve(pieces[p + b]) for b in range(len(Cs))])
            p += len(Cs)
        return out

    def forward(self, outputs, targets, mask_type, coords=None):
        final = {k: v for k, v in outputs.items() if k != "aux_outputs"}
        levels = [final] + list(outputs.get("aux_outputs", []))
        all_indices = self.match_all_levels(levels, targets, mask_type)

        num_masks = sum(len(t["labels"]) for t in targets)
        if is_dist_avail_and_initialized():     # the reference's only explicit collective (criterion.py:258-260)
            nm = torch.as_tensor([num_masks], dtype=torch.float, device=outputs["pred_logits"].device)
            torch.distributed.all_reduce(nm)
            num_masks = torch.clamp(nm / get_world_size(), min=1).item()
        else:
            num_masks = max(float(num_masks), 1.0)

        losses = {}
        for loss in self.losses:
            losses.update(self.get_loss(loss, final, targets, all_indices[0], num_masks, mask_type, coords))
        for i, aux in enumerate(levels[1:]):
            for loss in self.losses:
                ld = self.get_loss(loss, aux, targets, all_indices[i + 1], num_masks, mask_type, coords)
                losses.update({f"{k}_{i}": v for k, v in ld.items()})
        return losses
