"""oracle/sparse_ref.py — CPU restatement of the MinkowskiEngine-backed part of
UnScene3D's hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``unscene3d_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, as the checker.

PARITY UNPINNED for this file: the arithmetic it restates lives in
MinkowskiEngine 0.5.4 (NVIDIA/MinkowskiEngine, pinned by the reference at
conf/unscene3d_requirements.txt:50 and built from git master in
.devcontainer/Dockerfile:50-51), which is NOT vendored under /root/reference and
cannot be installed here.  The restatement follows ME's published semantics:

* ``sparse_quantize`` (called at reference datasets/utils.py:403-408): distinct
  coordinates in first-occurrence order; ``unique_map`` = first-occurrence row
  (ascending), ``inverse_map`` = row of the distinct coordinate
  (ME CoordinateMapCPU::insert_and_map, sequential insert).
* stride-2 coordinate maps: ``c -> floor(c / (2 ts)) * (2 ts)``, distinct, in
  first-occurrence order of the finer map's rows (ME CoordinateMapCPU::stride;
  GPU ME leaves the order implementation-defined — this build fixes it).
* kernel maps: HYPER_CUBE region, kernel 3 -> offsets {-1,0,1}^3 * ts with x
  fastest; kernel 2 / stride 2 -> offsets {0,1}^3 * ts, out = coarse map.
* conv: ``out[o] = sum_k sum_{(i,o) in M_k} in[i] @ W[k]``, W f32[K, Cin, Cout]
  (reference call sites models/modules/common.py:146,179);
  transposed conv uses the same map with in/out swapped.
* ``MinkowskiBatchNorm`` = ``torch.nn.BatchNorm1d`` over feature rows
  (models/modules/common.py:22); ``MinkowskiAvgPooling(2,2)`` = mean over
  present children (models/mask3d.py:131).

It is pinned only by the known-answer tests in tests/test_oracle_known_answers.py
(dense ``conv3d`` / ``conv_transpose3d`` on the densified grid, brute-force
unique / neighbour search).
"""
from __future__ import annotations

import numpy as np
import torch

COORD_BITS = 18
COORD_BIAS = 1 << (COORD_BITS - 1)


# --------------------------------------------------------------------------- V1
def voxel_floor(xyz: np.ndarray, voxel_size: float) -> np.ndarray:
    """np.floor(xyz / voxel_size) — reference datasets/utils.py:403."""
    return np.floor(np.asarray(xyz, dtype=np.float64) / voxel_size).astype(np.int32)


def pack_keys(coords: np.ndarray) -> np.ndarray:
    c = np.asarray(coords, dtype=np.int64)
    assert c.ndim == 2 and c.shape[1] == 4
    return (
        (c[:, 0] << (3 * COORD_BITS))
        | ((c[:, 1] + COORD_BIAS) << (2 * COORD_BITS))
        | ((c[:, 2] + COORD_BIAS) << COORD_BITS)
        | (c[:, 3] + COORD_BIAS)
    )


def quantize_coords(coords: np.ndarray, quant: int) -> np.ndarray:
    c = np.asarray(coords, dtype=np.int64).copy()
    if quant > 1:
        c[:, 1:] = np.floor_divide(c[:, 1:], quant) * quant
    return c.astype(np.int32)


def coordmap_build(coords: np.ndarray, quant: int = 1):
    """-> (unique_idx i64[n_out], inverse i64[n], out_coords i32[n_out,4]).

    First-occurrence unique (ME sparse_quantize / CoordinateMapCPU semantics)."""
    q = quantize_coords(coords, quant)
    if q.shape[0] == 0:
        return (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros((0, 4), np.int32))
    keys = pack_keys(q)
    _, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")  # distinct keys sorted by first occurrence
    rank = np.empty_like(order)
    rank[order] = np.arange(order.shape[0])
    unique_idx = first[order].astype(np.int64)
    inverse = rank[inv.reshape(-1)].astype(np.int64)
    return unique_idx, inverse, q[unique_idx]


def sparse_quantize(coordinates: np.ndarray):
    """ME.utils.sparse_quantize(coords, return_index=True, return_inverse=True) on
    [n,3] (single scene) or [n,4] coordinates -> (unique_map, inverse_map)."""
    c = np.asarray(coordinates)
    if c.shape[1] == 3:
        c = np.concatenate([np.zeros((c.shape[0], 1), c.dtype), c], axis=1)
    u, inv, _ = coordmap_build(c.astype(np.int32), 1)
    return u, inv


def sparse_collate(coords_list, feats_list):
    """ME.utils.sparse_collate: prepend batch index, concatenate (reference
    datasets/utils.py:430)."""
    cs, fs = [], []
    for b, (c, f) in enumerate(zip(coords_list, feats_list)):
        c = np.asarray(c, dtype=np.int32)
        cs.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], axis=1))
        fs.append(np.asarray(f, dtype=np.float32))
    return np.concatenate(cs, 0), np.concatenate(fs, 0)


# --------------------------------------------------------------------------- R2
def _lookup(sorted_keys, sorted_rows, query_keys):
    pos = np.searchsorted(sorted_keys, query_keys)
    pos = np.clip(pos, 0, len(sorted_keys) - 1)
    hit = sorted_keys[pos] == query_keys
    return np.where(hit, sorted_rows[pos], -1).astype(np.int32)


def cube_offsets(ksize: int = 3):
    r = ksize // 2
    offs = []
    for dz in range(-r, r + 1):
        for dy in range(-r, r + 1):
            for dx in range(-r, r + 1):
                offs.append((dx, dy, dz))
    return offs  # x fastest


def kernel_map_cube(coords: np.ndarray, tensor_stride: int, ksize: int = 3) -> np.ndarray:
    """Dense neighbour table nbr[K, n]: row with coord == coord_o + off_k*ts or -1."""
    coords = np.asarray(coords, dtype=np.int32)
    n = coords.shape[0]
    keys = pack_keys(coords)
    order = np.argsort(keys, kind="stable")
    sk, sr = keys[order], order.astype(np.int32)
    offs = cube_offsets(ksize)
    nbr = np.full((len(offs), n), -1, np.int32)
    for k, (dx, dy, dz) in enumerate(offs):
        q = coords.astype(np.int64).copy()
        q[:, 1] += dx * tensor_stride
        q[:, 2] += dy * tensor_stride
        q[:, 3] += dz * tensor_stride
        nbr[k] = _lookup(sk, sr, pack_keys(q)) if n else nbr[k]
    return nbr


def kernel_map_down2(fine_coords, tensor_stride, parent, coarse_coords):
    """Child table nbr2[8, n_coarse] and per-fine-row offset index kidx[u8]."""
    fine = np.asarray(fine_coords, dtype=np.int64)
    coarse = np.asarray(coarse_coords, dtype=np.int64)
    nc = coarse.shape[0]
    o = (fine[:, 1:] - coarse[parent][:, 1:]) // tensor_stride
    assert o.min(initial=0) >= 0 and o.max(initial=0) <= 1
    kidx = (o[:, 0] + 2 * o[:, 1] + 4 * o[:, 2]).astype(np.uint8)
    nbr2 = np.full((8, nc), -1, np.int32)
    nbr2[kidx, parent] = np.arange(fine.shape[0], dtype=np.int32)
    return nbr2, kidx


def rulebook_compact(nbr: np.ndarray):
    """[ME]-style per-offset pair lists ordered by (k, out row)."""
    K, n = nbr.shape
    ks, outs = np.nonzero(nbr >= 0)
    in_idx = nbr[ks, outs].astype(np.int32)
    out_idx = outs.astype(np.int32)
    koff = np.zeros(K + 1, np.int64)
    np.cumsum(np.bincount(ks, minlength=K), out=koff[1:])
    return in_idx, out_idx, koff


# --------------------------------------------------------------------------- C
def conv_gather(feats: torch.Tensor, W: torch.Tensor, nbr, n_out: int, bias=None) -> torch.Tensor:
    """out[o] = sum_k in[nbr[k,o]] @ W[k]; nbr None -> 1x1 (W [1,Cin,Cout] or [Cin,Cout])."""
    if W.dim() == 2:
        W = W[None]
    out = torch.zeros(n_out, W.shape[2], dtype=feats.dtype)
    if nbr is None:
        out = feats @ W[0]
    else:
        nbr_t = torch.as_tensor(np.asarray(nbr), dtype=torch.long)
        for k in range(W.shape[0]):
            rows = nbr_t[k]
            m = rows >= 0
            if m.any():
                out[m] += feats[rows[m]] @ W[k]
    if bias is not None:
        out = out + bias.reshape(1, -1)
    return out


def conv_transpose_up2(feats_coarse: torch.Tensor, W: torch.Tensor, parent, kidx, n_fine: int) -> torch.Tensor:
    """out[fine] = in[parent[fine]] @ W[kidx[fine]]  (MinkowskiConvolutionTranspose k=2,s=2)."""
    parent_t = torch.as_tensor(np.asarray(parent), dtype=torch.long)
    kidx_t = torch.as_tensor(np.asarray(kidx).astype(np.int64))
    out = torch.zeros(n_fine, W.shape[2], dtype=feats_coarse.dtype)
    for k in range(W.shape[0]):
        m = kidx_t == k
        if m.any():
            out[m] = feats_coarse[parent_t[m]] @ W[k]
    return out


# --------------------------------------------------------------------------- B, P, Q3
def batch_norm_train(x: torch.Tensor, gamma, beta, eps: float = 1e-5):
    """BatchNorm1d training forward over rows -> (y, mean, biased var)."""
    mean = x.mean(0)
    var = x.var(0, unbiased=False)
    y = (x - mean) / torch.sqrt(var + eps) * gamma + beta
    return y, mean, var


def avgpool_down2(feats: torch.Tensor, nbr2) -> torch.Tensor:
    nbr2_t = torch.as_tensor(np.asarray(nbr2), dtype=torch.long)
    nc = nbr2_t.shape[1]
    acc = torch.zeros(nc, feats.shape[1], dtype=feats.dtype)
    cnt = torch.zeros(nc, dtype=feats.dtype)
    for k in range(8):
        rows = nbr2_t[k]
        m = rows >= 0
        acc[m] += feats[rows[m]]
        cnt[m] += 1
    return acc / cnt.clamp(min=1)[:, None]


def scatter_mean(src: torch.Tensor, index: torch.Tensor, S: int | None = None) -> torch.Tensor:
    """torch_scatter.scatter_mean(src, index, dim=0) (reference models/mask3d.py:223)."""
    if S is None:
        S = int(index.max()) + 1
    out = torch.zeros(S, src.shape[1], dtype=src.dtype)
    out.index_add_(0, index, src)
    cnt = torch.bincount(index, minlength=S).clamp(min=1).to(src.dtype)
    return out / cnt[:, None]
