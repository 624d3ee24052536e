"""`torch_scatter.scatter_mean` surface (reference models/mask3d.py:12,223,
trainer/trainer.py:449): segment mean over dim 0 with forward + backward HIP kernels and a
stable counting-sort CSR (deterministic summation order)."""
import torch

from . import ops


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim: int = 0, dim_size: int = None, csr=None):
    if dim != 0 or src.dim() != 2:
        raise NotImplementedError("scatter_mean: only 2-D src reduced over dim 0 is on the hot path")
    if csr is None:
        S = int(dim_size) if dim_size is not None else int(index.max().item()) + 1
        csr = ops.segment_csr(index.to(torch.int64).contiguous(), S)
    return ops.segment_mean(src, csr)
