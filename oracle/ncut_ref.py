"""oracle/ncut_ref.py — CPU (numpy/torch/scipy) restatement of the masked-NCut functions of
pseudo_masks/unscene3d_pseudo_main.py:82-153,405-502 and utils/freemask_utils.py:8-18.
TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/ncut.npz, which was produced by importing the
reference's own `unscene3d()` in the build container (tests/golden/make_golden.py)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F
from scipy.linalg import eigh


def aggregate_features(feats, seg, conn, mode="mean"):
    """N1 (reference pseudo_masks/unscene3d_pseudo_main.py:350-402): per-segment mean (or max) over the rows that have
    a non-zero entry (:362-366); a segment without one gets the mean of the non-zero aggregated features of the
    segments connected to `zero_segments[0]` — the FIRST zero segment's neighbours for every zero segment (:387) — or
    the mean over all aggregated rows when none qualifies (:397), each fill visible to the next one.
    numpy f32 in torch's reduction conventions (pairwise-summed means), -> (f32[S,d], unique segment ids)."""
    feats = np.asarray(feats, np.float32)
    seg = np.asarray(seg)
    conn = np.asarray(conn)
    uniq = np.unique(seg)
    valid = np.any(feats != 0, axis=-1)
    agg = np.zeros((len(uniq), feats.shape[1]), np.float32)
    for i, s in enumerate(uniq):
        rows = feats[valid & (seg == s)]
        if len(rows):
            agg[i] = rows.max(0) if mode == "max" else rows.mean(0, dtype=np.float32)
    zero = uniq[np.all(agg == 0, axis=-1)]
    if len(zero):
        nbr = conn[conn[:, 0] == zero[0]][:, 1]
        nbr_idx = np.array([int(np.nonzero(uniq == s)[0][0]) for s in nbr], dtype=np.int64)
        for z in zero:
            cand = agg[nbr_idx] if len(nbr_idx) else agg[:0]
            cand = cand[np.any(cand != 0, axis=-1)]
            agg[int(np.nonzero(uniq == z)[0][0])] = (cand if len(cand) else agg).mean(0, dtype=np.float32)
    return agg, uniq


def cosine_sim(k, q):
    eps = 10e-10
    kf = k / (k.norm(dim=1, keepdim=True) + eps)
    qf = q / (q.norm(dim=1, keepdim=True) + eps)
    attn = qf @ kf.T
    attn -= attn.min(-1, keepdim=True)[0]
    attn /= attn.max(-1, keepdim=True)[0] + eps
    return attn


def normalize_mat(A, eps=1e-5):
    A -= np.min(A[np.nonzero(A)]) if np.any(A > 0) else 0
    A[A < 0] = 0.0
    A /= A.max() + eps
    return A


def affinity(feats, tau, eps=1e-5):
    """-> (A f64[S,S] in {1, eps}, d f64[S]) (reference get_affinity_matrix :89-119)."""
    if not isinstance(feats, tuple):
        fa = F.normalize(feats, p=2, dim=-1)
        A = normalize_mat(cosine_sim(fa, fa).numpy())
    else:
        mats = []
        for f in feats:
            fn = F.normalize(f, p=2, dim=-1)
            mats.append(normalize_mat((fn @ fn.T).numpy()))
        A = (mats[0] + mats[1]) / 2
    A = A > tau
    A = np.where(A.astype(float) == 0, eps, A)
    return A, np.sum(A, axis=0)


def fiedler(A, d):
    w, v = eigh(np.diag(d) - A, np.diag(d), subset_by_index=[1, 2])
    return w, v[:, 0]


def _separate(bipartition, vec, uniq, conn):
    neighbours = {int(s): set(conn[conn[:, 0] == s, 1].tolist()) for s in uniq}
    blobs = []
    for c in uniq[bipartition]:
        nb = neighbours[int(c)]
        last, merged, k = -1, False, 0
        while k < len(blobs):
            if nb & blobs[k]:
                merged = True
                blobs[k].add(int(c))
                if last != -1:
                    blobs[last] = blobs[last] | blobs[k]
                    blobs.pop(k)
                else:
                    last = k
            k += 1
        if not merged:
            blobs.append({int(c)})
    seed = int(uniq[int(np.argmax(vec))])
    return next(b for b in blobs if seed in b)


def unscene3d_ref(feats, uniq, conn, tau=0.6, max_instances=20, max_extent_ratio=0.8, eps=1e-5, min_segment_size=4,
                  trace=None):
    """CPU restatement of unscene3d() (reference :405-502, separation_mode='max', visualisation dropped).
    feats: torch f32[S,d] or a 2-tuple; uniq i64[S]; conn i64[E,2].  -> bool[K,S]"""
    uniq = np.asarray(uniq)
    conn = np.asarray(conn)
    S = len(uniq)
    painting = torch.zeros(S)
    out, fg, current = [], set(), None
    for it in range(max_instances):
        if it > 0:
            p = ((painting.view(S, 1) + current.view(S, 1).float()) > 0).float()
            feats = tuple((1 - p) * f for f in feats) if isinstance(feats, tuple) else (1 - p) * feats
            painting = p.squeeze()
        A, d = affinity(tuple(f.clone() for f in feats) if isinstance(feats, tuple) else feats.clone(), tau, eps)
        pb = painting.bool().numpy()
        A[pb] = eps
        A[:, pb] = eps
        w, vec = fiedler(A, d)
        bip = vec > np.sum(vec) / len(vec)
        if bip.sum() / len(bip) > max_extent_ratio:
            bip, vec = np.logical_not(bip), -vec
        part = _separate(bip, vec, uniq, conn)
        current = torch.as_tensor(np.isin(uniq, list(part)))
        if trace is not None:
            trace.append({"it": it, "evals": w, "n_fg": int(bip.sum()), "part": sorted(part), "A_on": int((A > 0.5).sum())})
        if len(part & fg) / len(part) > 0.5 or len(part) < min_segment_size:
            continue
        out.append(np.isin(uniq, list(part - fg)))
        fg |= part
    return np.stack(out) if out else np.zeros((0, S), dtype=bool)
