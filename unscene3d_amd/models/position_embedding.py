"""PositionEmbeddingCoordsSine (reference models/position_embedding.py:43-179).

'fourier' (the hot-path setting, conf/model/mask3d.yaml:14): random Gaussian projection of the
min-max normalised coordinates, `[sin | cos](2*pi * xn @ gauss_B)`; the buffer `gauss_B` f32[3, d/2]
is created at construction and lives in checkpoints.  The projection + sin/cos is one HIP kernel."""
import math

import torch
from torch import nn

from .. import ops


def shift_scale_points(pred_xyz, src_range, dst_range=None):
    """Affine map of [B,N,3] points from src_range=[min,max] ([B,3] each) to dst_range (default [0,1])."""
    lo, hi = src_range
    if dst_range is None:
        dlo, dhi = torch.zeros_like(lo), torch.ones_like(hi)
    else:
        dlo, dhi = dst_range
    return ((pred_xyz - lo[:, None, :]) * (dhi - dlo)[:, None, :]) / (hi - lo)[:, None, :] + dlo[:, None, :]


class PositionEmbeddingCoordsSine(nn.Module):
    def __init__(self, temperature=10000, normalize=False, scale=None, pos_type="fourier", d_pos=None, d_in=3,
                 gauss_scale=1.0):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        assert pos_type in ("sine", "fourier")
        self.d_pos, self.temperature, self.normalize = d_pos, temperature, normalize
        self.pos_type = pos_type
        self.scale = 2 * math.pi if scale is None else scale
        if pos_type == "fourier":
            assert d_pos is not None and d_pos % 2 == 0
            self.register_buffer("gauss_B", torch.empty((d_in, d_pos // 2)).normal_() * gauss_scale)

    # -- fast path used by Mask3D: one scene, rows out
    def fourier_rows(self, xyz: torch.Tensor, lo: torch.Tensor, hi: torch.Tensor) -> torch.Tensor:
        """xyz f32[N,3], lo/hi f32[3] -> f32[N, d_pos] (= forward(...)[0].T of the reference)."""
        if not self.normalize:
            lo, hi = torch.zeros_like(lo), torch.ones_like(hi)
        return ops.fourier_posenc(xyz.float().contiguous(), lo.float(), hi.float(), self.gauss_B, self.d_pos)

    def get_fourier_embeddings(self, xyz, num_channels=None, input_range=None):
        if num_channels is not None and num_channels != self.d_pos:
            raise NotImplementedError("num_channels != d_pos is not used by the hot path")
        B = xyz.shape[0]
        rows = [self.fourier_rows(xyz[b], input_range[0][b], input_range[1][b]) if self.normalize else
                self.fourier_rows(xyz[b], xyz.new_zeros(3), xyz.new_ones(3)) for b in range(B)]
        return torch.stack(rows).permute(0, 2, 1)          # batch x d_pos x npoints

    def get_sine_embeddings(self, xyz, num_channels, input_range):
        xyz = xyz.clone()
        if self.normalize:
            xyz = shift_scale_points(xyz, src_range=input_range)
        d_in = xyz.shape[2]
        ndim = self.d_pos // d_in
        ndim -= ndim % 2
        rems = self.d_pos - ndim * d_in
        parts = []
        for d in range(d_in):
            cdim = ndim + (2 if rems > 0 else 0)
            rems -= 2 if rems > 0 else 0
            t = torch.arange(cdim, dtype=torch.float32, device=xyz.device)
            t = self.temperature ** (2 * torch.div(t, 2, rounding_mode="floor") / cdim)
            pos = (xyz[:, :, d] * self.scale)[:, :, None] / t
            parts.append(torch.stack((pos[:, :, 0::2].sin(), pos[:, :, 1::2].cos()), dim=3).flatten(2))
        return torch.cat(parts, dim=2).permute(0, 2, 1)

    def forward(self, xyz, num_channels=None, input_range=None):
        assert isinstance(xyz, torch.Tensor) and xyz.ndim == 3
        with torch.no_grad():
            if self.pos_type == "fourier":
                return self.get_fourier_embeddings(xyz, num_channels, input_range)
            return self.get_sine_embeddings(xyz, num_channels, input_range)
