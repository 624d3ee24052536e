"""Data-parallel gradient exchange: one process per GPU, one scene per
rank, gradients all-reduced over RCCL/xGMI.  The reference gets this implicitly from
PyTorch-Lightning DDP (main_instance_segmentation.py:86-92); here every parameter's
`.grad` is a view into ONE flat fp32 buffer so the exchange is a single large
all-reduce (xGMI rings are per-link bound: few, large collectives)."""
from __future__ import annotations

import torch


def flatten_grads(params):
    """Allocate one flat buffer and make each p.grad a view into it. Returns the buffer."""
    params = list(params)
    total = sum(p.numel() for p in params)
    flat = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
    off = 0
    for p in params:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p)
        off += n
    return flat


def all_reduce_mean_(flat, world_size: int):
    import torch.distributed as dist

    if world_size > 1:
        dist.all_reduce(flat)
        flat.div_(world_size)
    return flat
