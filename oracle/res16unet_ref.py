"""oracle/res16unet_ref.py — CPU (torch autograd) restatement of the Res16UNet
forward used by the hot path (reference models/res16unet.py:224-297 topology,
models/modules/resnet_block.py:48-64 block, models/resnet.py:96-149 downsample).

TEST INFRASTRUCTURE ONLY (see oracle/sparse_ref.py header: parity unpinned for the
MinkowskiEngine-backed arithmetic).  It consumes the *state_dict* of the device
model, so parameter names double as the checkpoint-key contract.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import sparse_ref as R


class Pyramid:
    """Coordinate maps + kernel maps of one batch, built with the oracle."""

    def __init__(self, coords: np.ndarray, levels: int = 5):
        self.coords = [np.asarray(coords, np.int32)]
        self.parent, self.nbr2, self.kidx, self.cube = [], [], [], {}
        ts = 1
        for _ in range(levels - 1):
            _, parent, cc = R.coordmap_build(self.coords[-1], 2 * ts)
            nbr2, kidx = R.kernel_map_down2(self.coords[-1], ts, parent, cc)
            self.parent.append(parent)
            self.nbr2.append(nbr2)
            self.kidx.append(kidx)
            self.coords.append(cc)
            ts *= 2

    def cube_map(self, level: int):
        if level not in self.cube:
            self.cube[level] = R.kernel_map_cube(self.coords[level], 1 << level)
        return self.cube[level]


# Index lists of a kernel map, built once per table object: per offset k the output rows that have a neighbour and
# the input rows they read.  (The lists were rebuilt — and the [N, C] accumulator copied — for every offset of every
# call: 20x the time of the 27 matrix products themselves at 150 k voxels; same products, same order, same bits.)
_PAIR_CACHE = {}


def _pairs_of(table, build, also=None):
    ent = _PAIR_CACHE.get(id(table))
    if ent is None or ent[0] is not table or ent[2] is not also:
        if len(_PAIR_CACHE) >= 64:
            _PAIR_CACHE.clear()
        ent = (table, build(), also)     # (the table is kept alive with its lists: an id is never re-used under us)
        _PAIR_CACHE[id(table)] = ent
    return ent[1]


def _gather_conv(x, W, nbr, n_out):
    def build():
        nbr_t = torch.as_tensor(np.asarray(nbr), dtype=torch.long)
        out = []
        for k in range(nbr_t.shape[0]):
            rows = nbr_t[k]
            m = torch.nonzero(rows >= 0).reshape(-1)
            out.append((m, rows[m]))
        return out
    pairs = _pairs_of(nbr, build)
    out = torch.zeros(n_out, W.shape[2], dtype=x.dtype)
    for k in range(W.shape[0]):            # offsets ascending: the summation order of an output row
        m, src = pairs[k]
        if m.numel():
            out.index_add_(0, m, x[src] @ W[k])
    return out


def _tr_conv(x, W, parent, kidx, n_fine):
    def build():
        parent_t = torch.as_tensor(np.asarray(parent), dtype=torch.long)
        kidx_t = torch.as_tensor(np.asarray(kidx).astype(np.int64))
        out = []
        for k in range(int(kidx_t.max()) + 1 if kidx_t.numel() else 0):
            m = torch.nonzero(kidx_t == k).reshape(-1)
            out.append((m, parent_t[m]))
        return out
    pairs = _pairs_of(kidx, build, also=parent)
    out = torch.zeros(n_fine, W.shape[2], dtype=x.dtype)
    for k in range(min(W.shape[0], len(pairs))):
        m, src = pairs[k]
        if m.numel():
            out.index_add_(0, m, x[src] @ W[k])
    return out


_EVAL = [False]     # set by res16unet_forward(is_eval=...): MinkowskiBatchNorm in eval() uses its running statistics


def _bn(sd, name, x, eps=1e-5):
    """MinkowskiBatchNorm = nn.BatchNorm1d over all rows of the batch (models/modules/common.py:22).  Training mode:
    batch statistics; eval mode (`module.eval()`, trainer/trainer.py:384-396 runs under it): the running mean / variance
    of the state_dict."""
    if _EVAL[0]:
        rm, rv = sd[name + ".bn.running_mean"].to(x.dtype), sd[name + ".bn.running_var"].to(x.dtype)
        return F.batch_norm(x, rm, rv, sd[name + ".bn.weight"], sd[name + ".bn.bias"], training=False, eps=eps)
    return F.batch_norm(x, None, None, sd[name + ".bn.weight"], sd[name + ".bn.bias"], training=True, eps=eps)


def _block(sd, prefix, x, pyr, level):
    nbr = pyr.cube_map(level)
    n = x.shape[0]
    out = torch.relu(_bn(sd, prefix + ".norm1", _gather_conv(x, sd[prefix + ".conv1.kernel"], nbr, n)))
    out = _bn(sd, prefix + ".norm2", _gather_conv(out, sd[prefix + ".conv2.kernel"], nbr, n))
    res = x
    if prefix + ".downsample.0.kernel" in sd:
        res = _bn(sd, prefix + ".downsample.1", x @ sd[prefix + ".downsample.0.kernel"])
    return torch.relu(out + res)


def _layer(sd, name, x, pyr, level, nblocks):
    for b in range(nblocks):
        x = _block(sd, f"{name}.{b}", x, pyr, level)
    return x


def res16unet_forward(sd: dict, pyr: Pyramid, feats: torch.Tensor, layers, is_eval: bool = False):
    """-> (stride-1 features, [s16, s8, s4, s2, s1] block outputs).  is_eval: batch norms use running statistics."""
    _EVAL[0] = bool(is_eval)
    try:
        return _forward(sd, pyr, feats, layers)
    finally:
        _EVAL[0] = False


def _forward(sd, pyr, feats, layers):
    n0 = feats.shape[0]
    x = torch.relu(_bn(sd, "bn0", _gather_conv(feats, sd["conv0p1s1.kernel"], pyr.cube_map(0), n0)))
    skips = [x]
    down = ("conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2")
    for i, cname in enumerate(down):
        nc = pyr.coords[i + 1].shape[0]
        x = torch.relu(_bn(sd, f"bn{i + 1}", _gather_conv(x, sd[cname + ".kernel"], pyr.nbr2[i], nc)))
        x = _layer(sd, f"block{i + 1}", x, pyr, i + 1, layers[i])
        skips.append(x)
    levels = [x]
    up = ("convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2")
    for j, cname in enumerate(up):
        fine = 3 - j
        nf = pyr.coords[fine].shape[0]
        x = _tr_conv(x, sd[cname + ".kernel"], pyr.parent[fine], pyr.kidx[fine], nf)
        x = torch.relu(_bn(sd, f"bntr{4 + j}", x))
        x = torch.cat([x, skips[fine]], dim=1)
        x = _layer(sd, f"block{5 + j}", x, pyr, fine, layers[4 + j])
        levels.append(x)
    return x, levels
