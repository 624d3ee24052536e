"""`custom_cuda_utils` surface used by the hot path (reference utils/cuda_utils/cuda_utils.cpp:26-46):
in-place fill of caller-allocated, zeroed outputs; returns None — like the reference."""
import torch

from ._lib import check, lib
from .ops import _chk, _ptr, _stream


def project_sparse_voxels_to_planes(s_coords, s_predictions, s_targets, xy_pred, xz_pred, yz_pred, xy_tgt, xz_tgt,
                                    yz_tgt, xy_nums, xz_nums, yz_nums):
    _chk(s_coords, torch.int32, "s_coords")
    _chk(s_predictions, torch.float32, "s_predictions")
    _chk(s_targets, torch.float32, "s_targets")
    V, inst = s_predictions.shape
    x_dim, y_dim = xy_nums.shape
    z_dim = xz_nums.shape[1]
    check(lib.usc_project_planes_fwd(_ptr(s_coords), _ptr(s_predictions), _ptr(s_targets), V, inst, x_dim, y_dim, z_dim,
                                     _ptr(xy_pred), _ptr(xz_pred), _ptr(yz_pred), _ptr(xy_tgt), _ptr(xz_tgt),
                                     _ptr(yz_tgt), _ptr(xy_nums), _ptr(xz_nums), _ptr(yz_nums), _stream()),
          "usc_project_planes_fwd")


def project_sparse_voxels_to_planes_backward(s_coords, s_grads, xy_grads, xz_grads, yz_grads, xy_nums, xz_nums,
                                             yz_nums):
    V, inst = s_grads.shape
    x_dim, y_dim = xy_nums.shape
    z_dim = xz_nums.shape[1]
    check(lib.usc_project_planes_bwd(_ptr(s_coords), V, inst, x_dim, y_dim, z_dim, _ptr(xy_grads), _ptr(xz_grads),
                                     _ptr(yz_grads), _ptr(s_grads), _stream()), "usc_project_planes_bwd")
