"""Evaluation-time mask splitting (reference trainer/trainer.py:507-539, `general.use_dbscan`): every query
mask is split into the connected components of its eps-ball graph.  The reference calls
sklearn DBSCAN(eps=0.95, min_samples=1) once per query; here the components come from the device
(`ops.cc_eps`, identical labels)."""
from __future__ import annotations

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
    return torch.stack(new_masks).T, torch.stack(new_logits)
