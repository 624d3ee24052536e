"""Tri-plane projection loss (reference models/noise_robust_loss.py:16-163): sigmoid mask predictions and
targets of a scene are averaged along z / y / x into XY / XZ / YZ images and compared with BCE.
Active only when `matcher.cost_noise_robust != 0` (default 0, conf/matcher/hungarian_matcher.yaml:6).
Projection forward/backward are HIP kernels behind the reference's `custom_cuda_utils` interface."""
import torch
from torch import nn
from torch.autograd import Function

from .. import custom_cuda_utils


class ProjectionFunction(Function):
    @staticmethod
    def forward(ctx, s_coords, s_predictions, s_targets, xy_pred, xz_pred, yz_pred, xy_tgt, xz_tgt, yz_tgt, xy_nums,
                xz_nums, yz_nums):
        s_grads = torch.zeros(s_predictions.shape, device=s_coords.device)
        custom_cuda_utils.project_sparse_voxels_to_planes(s_coords, s_predictions, s_targets, xy_pred, xz_pred, yz_pred,
                                                          xy_tgt, xz_tgt, yz_tgt, xy_nums, xz_nums, yz_nums)
        ctx.save_for_backward(s_coords, s_grads, xy_nums, xz_nums, yz_nums)
        outs = []
        for plane, nums in ((xy_pred, xy_nums), (xz_pred, xz_nums), (yz_pred, yz_nums), (xy_tgt, xy_nums),
                            (xz_tgt, xz_nums), (yz_tgt, yz_nums)):
            p = plane / (nums.unsqueeze(-1) + 10e-9)       # mean over the voxels that hit the pixel
            p[nums == 0] = 0.0
            outs.append(p)
        return (*outs, xy_nums, xz_nums, yz_nums)

    @staticmethod
    def backward(ctx, g_xy, g_xz, g_yz, *unused):
        s_coords, s_grads, xy_nums, xz_nums, yz_nums = ctx.saved_tensors
        # like the reference, the plane gradients are handed to the voxels un-normalised (:62-68)
        custom_cuda_utils.project_sparse_voxels_to_planes_backward(s_coords, s_grads, g_xy.contiguous(),
                                                                   g_xz.contiguous(), g_yz.contiguous(), xy_nums,
                                                                   xz_nums, yz_nums)
        return (None, s_grads) + (None,) * 10


class ProjectionFunctionWrapper(nn.Module):
    def forward(self, s_coords, s_predictions, s_targets):
        centered = s_coords - torch.amin(s_coords, 0)
        x_dim, y_dim, z_dim = (int(v) for v in centered[:, 1:].max(0)[0])   # max, not max+1: reference :80
        n = s_predictions.shape[1]
        dev = s_coords.device
        planes = [torch.zeros((a, b, n), device=dev) for a, b in ((x_dim, y_dim), (x_dim, z_dim), (y_dim, z_dim))] * 2
        planes = [p.clone() for p in planes]
        nums = [torch.zeros((a, b), device=dev, dtype=torch.int) for a, b in
                ((x_dim, y_dim), (x_dim, z_dim), (y_dim, z_dim))]
        out = ProjectionFunction.apply(centered.int().contiguous(), s_predictions.contiguous(),
                                       s_targets.contiguous(), *planes, *nums)
        return out, (x_dim, y_dim, z_dim)


class ProjectionMaskLoss(nn.Module):
    def __init__(self, config=None, base_loss="bce", directions="xyz"):
        super().__init__()
        if base_loss != "bce":
            raise NotImplementedError
        self.base_loss, self.eps, self.directions = base_loss, 10e-9, directions
        self.projection_module = ProjectionFunctionWrapper()
        self.criterion = nn.BCELoss(reduction="none")

    def forward(self, all_mask_preds, all_mask_targets, coords):
        inst_num, _ = all_mask_preds.shape
        outs, _ = self.projection_module(coords, torch.sigmoid(all_mask_preds.T), all_mask_targets.T)
        xy_p, xz_p, yz_p, xy_t, xz_t, yz_t, xy_n, xz_n, yz_n = outs
        all_shape = inst_num * (len(xy_n.nonzero()) + len(xz_n.nonzero()) + len(yz_n.nonzero()))
        loss = 0
        for key, pred, tgt, nums in (("x", yz_p, yz_t, yz_n), ("y", xz_p, xz_t, xz_n), ("z", xy_p, xy_t, xy_n)):
            if key in self.directions:
                l = self.criterion(pred, tgt.detach())
                l[nums == 0] = 0.0
                loss = loss + l.sum()
        return loss, all_shape
