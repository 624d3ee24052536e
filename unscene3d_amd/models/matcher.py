"""HungarianMatcher (reference models/matcher.py:67-201).

Cost = cost_mask * BCE + cost_class * (-p_class) + cost_dice * dice between every query and every
target mask (two `nc,mc->nm` contractions on [Q, S] x [T, S]); the linear-sum-assignment itself is
scipy's on the host, exactly like the reference (:161-163).  `cost_matrices()` exposes the device
part so that SetCriterion can build the matrices of all 13 prediction levels first and pay ONE
device->host copy per step instead of 13*B."""
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment
from torch import nn


def batch_dice_loss(inputs: torch.Tensor, targets: torch.Tensor):
    """inputs [Q,S] logits, targets [T,S] in {0,1} -> [Q,T] dice cost."""
    p = inputs.sigmoid().flatten(1)
    num = 2 * (p @ targets.T)
    den = p.sum(-1)[:, None] + targets.sum(-1)[None, :]
    return 1 - (num + 1) / (den + 1)


def batch_sigmoid_ce_loss(inputs: torch.Tensor, targets: torch.Tensor):
    """[Q,S] logits vs [T,S] targets -> [Q,T] mean BCE cost."""
    hw = inputs.shape[1]
    pos = F.binary_cross_entropy_with_logits(inputs, torch.ones_like(inputs), reduction="none")
    neg = F.binary_cross_entropy_with_logits(inputs, torch.zeros_like(inputs), reduction="none")
    return (pos @ targets.T + neg @ (1 - targets).T) / hw


batch_dice_loss_jit = batch_dice_loss
batch_sigmoid_ce_loss_jit = batch_sigmoid_ce_loss


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_mask: float = 1, cost_dice: float = 1,
                 cost_noise_robust: float = 1.0, num_points: int = 0):
        super().__init__()
        self.cost_class, self.cost_mask, self.cost_dice = cost_class, cost_mask, cost_dice
        self.cost_noise_robust = cost_noise_robust
        if self.cost_class == 0 and self.cost_mask == 0 and self.cost_dice == 0:
            self.cost_mask = 1
        self.num_points = num_points

    @torch.no_grad()
    def cost_matrices(self, outputs, targets, mask_type):
        """-> list (per scene) of [Q, T_b] float32 cost matrices on the device."""
        Cs = []
        bs, num_queries = outputs["pred_logits"].shape[:2]
        for b in range(bs):
            out_prob = outputs["pred_logits"][b].softmax(-1)
            tgt_ids = targets[b]["labels"].clone()
            ignore = tgt_ids == 253
            tgt_ids[ignore] = 0
            cost_class = -out_prob[:, tgt_ids]
            cost_class[:, ignore] = -1.0
            out_mask = outputs["pred_masks"][b].T.float()                    # [Q, S]
            tgt_mask = targets[b][mask_type].to(out_mask)                     # [T, S]
            if self.num_points != -1:
                point_idx = torch.randperm(tgt_mask.shape[1], device=tgt_mask.device)[
                    :int(self.num_points * tgt_mask.shape[1])]
                out_mask, tgt_mask = out_mask[:, point_idx], tgt_mask[:, point_idx]
            C = (self.cost_mask * batch_sigmoid_ce_loss(out_mask, tgt_mask)
                 + self.cost_class * cost_class
                 + self.cost_dice * batch_dice_loss(out_mask, tgt_mask))
            Cs.append(C.reshape(num_queries, -1))
        return Cs

    @staticmethod
    def solve(C_cpu):
        i, j = linear_sum_assignment(C_cpu)
        return torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)

    @torch.no_grad()
    def memory_efficient_forward(self, outputs, targets, mask_type):
        return [self.solve(C.cpu()) for C in self.cost_matrices(outputs, targets, mask_type)]

    @torch.no_grad()
    def forward(self, outputs, targets, mask_type):
        return self.memory_efficient_forward(outputs, targets, mask_type)

    def __repr__(self, _repr_indent=4):
        pad = " " * _repr_indent
        return "\n".join(["Matcher " + self.__class__.__name__, f"{pad}cost_class: {self.cost_class}",
                          f"{pad}cost_mask: {self.cost_mask}", f"{pad}cost_dice: {self.cost_dice}"])
