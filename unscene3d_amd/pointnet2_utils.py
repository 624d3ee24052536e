"""`third_party.pointnet2.pointnet2_utils` surface used by the hot path: only
`furthest_point_sample` (reference models/mask3d.py:10,228;
third_party/pointnet2/pointnet2_utils.py:50-79).  Non-differentiable, like the reference."""
import torch

from . import ops


def furthest_point_sample(xyz: torch.Tensor, npoint: int) -> torch.Tensor:
    """xyz f32[B,N,3] contiguous on the HIP device -> i32[B,npoint] indices (first index 0)."""
    with torch.no_grad():
        return ops.furthest_point_sample(xyz.contiguous(), int(npoint))
