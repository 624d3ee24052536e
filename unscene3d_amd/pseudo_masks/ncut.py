"""Iterative masked Normalized Cut over segment features — the pseudo-mask generator's core
(reference pseudo_masks/unscene3d_pseudo_main.py: normalize_mat :82-86, get_affinity_matrix :89-119,
get_masked_affinity_matrix :122-135, second_smallest_eigenvector :138-146, get_salient_areas :149-153,
separate_segments :181-250, segment_ids_to_mask :254-260, aggregate_features :350-402, unscene3d :405-502).

Same function names and argument meaning.  The S x S work (similarity, normalisation, thresholding,
degree, generalized eigenvector) runs on the MI355X through libusc3d_hip.so; the S-sized set logic of
the cut (bipartition, flip rule, connectivity split around the arg-max seed, IoU / size gates) stays
on the host exactly like the reference — it is bookkeeping over a few hundred segment ids.
The visualisation-only accumulations of the reference loop (:441-447, :478-489) are not reproduced.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from .._lib import check, lib

max_number_of_instances = 20


def _similarity(feats: torch.Tensor, cosine_mode: bool, zero_rows=None) -> torch.Tensor:
    feats = feats.float().contiguous()
    S, d = feats.shape
    normed = torch.empty_like(feats)
    sim = torch.empty((S, S), dtype=torch.float32, device=feats.device)
    check(lib.usc_ncut_similarity_masked(feats.data_ptr(), None if zero_rows is None else zero_rows.data_ptr(), S, d,
                                         int(cosine_mode), normed.data_ptr(), sim.data_ptr(), ops._stream()),
          "usc_ncut_similarity_masked")
    return sim


def _l2_similarity(feats: torch.Tensor, zero_rows=None, block: int = 256) -> torch.Tensor:
    """`l2_sim(F.normalize(feats), F.normalize(feats))` of the reference (utils/freemask_utils.py:20-36; selected by
    similarity_metric='l2', unscene3d_pseudo_main.py:95): pairwise Euclidean distances of the L2-normalised rows, per-row
    min-max normalisation (no epsilon, as the reference), 1 - distance.  Not used by the shipped configurations
    (pseudo_masks/config/default.yaml: cos): plain device tensor ops in the reference's operation order, keys in blocks."""
    f = feats.float()
    if zero_rows is not None:
        f = f * (1.0 - zero_rows.to(f.dtype))[:, None]       # get_masked_affinity_matrix's product (:122-135)
    f = torch.nn.functional.normalize(f, p=2, dim=-1)
    S = f.shape[0]
    attn = torch.empty((S, S), dtype=torch.float32, device=f.device)
    for k0 in range(0, S, block):
        attn[:, k0:k0 + block] = torch.linalg.norm(f[:, None, :] - f[None, k0:k0 + block, :], dim=-1)
    attn -= attn.min(-1, keepdim=True)[0]
    attn /= attn.max(-1, keepdim=True)[0]
    return (1.0 - attn).contiguous()


def normalize_mat(A: torch.Tensor, eps=1e-5) -> torch.Tensor:
    """In place (reference :82-86), on the device."""
    ws = torch.empty(12288, dtype=torch.uint8, device=A.device)
    check(lib.usc_ncut_normalize_mat(A.data_ptr(), A.shape[0], ws.data_ptr(), ws.numel(), ops._stream()),
          "usc_ncut_normalize_mat")
    return A


def get_affinity_matrix(feats, tau=0.15, eps=1e-5, normalize_sim=True, similarity_metric="cos", painting=None,
                        zero_rows=None):
    """-> (A u8[S,S] on the device with A_ij = 1 meaning affinity 1 and 0 meaning `eps`, deg f64[S]).

    Single modality: row-min-max normalised cosine similarity; tuple of two modalities: plain normalised
    Gram matrices, each passed through normalize_mat, averaged.  `painting` (bool[S]) applies the
    reference's `A[painting] = eps; A[:, painting] = eps` (unscene3d :426-427) in the same launch.
    `zero_rows` (device u8[S]): rows of `feats` to read as `0 * row` — get_masked_affinity_matrix's product formed
    inside the row normalisation (the cut loop passes the ORIGINAL features and the painting so far)."""
    if similarity_metric not in ("cos", "l2"):
        raise ValueError(f"similarity_metric must be 'cos' or 'l2', not {similarity_metric!r}")
    if similarity_metric == "l2" and not isinstance(feats, tuple):
        sims = [_l2_similarity(feats, zero_rows)]           # reference :95 -> utils/freemask_utils.py:20-36
    elif isinstance(feats, tuple):
        sims = [_similarity(f, cosine_mode=False, zero_rows=zero_rows) for f in feats]
    else:
        sims = [_similarity(feats, cosine_mode=True, zero_rows=zero_rows)]
    if normalize_sim:
        for sm in sims:
            normalize_mat(sm)
    S = sims[0].shape[0]
    dev = sims[0].device
    A = torch.empty((S, S), dtype=torch.uint8, device=dev)
    deg = torch.empty(S, dtype=torch.float64, device=dev)
    if painting is None:
        pb = None
    elif painting.dtype == torch.uint8 and painting.device == dev and painting.is_contiguous():
        pb = painting
    else:
        pb = painting.to(device=dev, dtype=torch.uint8).contiguous()
    check(lib.usc_ncut_binarize(sims[0].data_ptr(), sims[1].data_ptr() if len(sims) > 1 else None, S, float(tau),
                                float(eps), None if pb is None else pb.data_ptr(), A.data_ptr(), deg.data_ptr(),
                                ops._stream()), "usc_ncut_binarize")
    return A, deg


def get_masked_affinity_matrix(painting, feats, mask):
    """Zero the features of painted segments (reference :122-135)."""
    S = feats[0].shape[0] if isinstance(feats, tuple) else feats.shape[0]
    painting = torch.logical_or(painting.view(S, 1) > 0, mask.view(S, 1) > 0).float()
    keep = 1 - painting                       # exactly 0 or 1: (1 - painting) * f as in the reference
    if isinstance(feats, tuple):
        feats = tuple(keep * f for f in feats)
    else:
        feats = keep * feats
    return feats, painting.squeeze()


def second_smallest_eigenvector(A, D, eps=1e-5):
    """Generalized eigenvector #2 of (D - A, D) — `eigh(D - A, D, subset_by_index=[1, 2])` of the reference
    (:138-146), computed on the device with LAPACK's sign.  A: device u8 affinity, D: device degree vector."""
    S = A.shape[0]
    evec = torch.empty(S, dtype=torch.float64, device=A.device)
    evals = torch.empty(2, dtype=torch.float64, device=A.device)
    ws = torch.empty(lib.usc_ncut_fiedler_ws_bytes(S), dtype=torch.uint8, device=A.device)
    check(lib.usc_ncut_fiedler(A.data_ptr(), D.data_ptr(), S, float(eps), evec.data_ptr(), evals.data_ptr(),
                               ws.data_ptr(), ws.numel(), ops._stream()), "usc_ncut_fiedler")
    vec = evec.cpu().numpy()
    return np.copy(vec), vec


def second_smallest_eigenvector_async(A, D, eps=1e-5, host=None):
    """The same solve without waiting for it: -> (pinned host f64[S], event).  The vector is valid once the event has
    fired (`event.synchronize()`); nothing else on the host waits, so several scenes' solves can be in flight on
    their own streams from ONE host thread (`unscene3d_steps`)."""
    S = A.shape[0]
    evec = torch.empty(S, dtype=torch.float64, device=A.device)
    evals = torch.empty(2, dtype=torch.float64, device=A.device)
    ws = torch.empty(lib.usc_ncut_fiedler_ws_bytes(S), dtype=torch.uint8, device=A.device)
    check(lib.usc_ncut_fiedler(A.data_ptr(), D.data_ptr(), S, float(eps), evec.data_ptr(), evals.data_ptr(),
                               ws.data_ptr(), ws.numel(), ops._stream()), "usc_ncut_fiedler")
    if host is None or host.numel() != S:
        host = torch.empty(S, dtype=torch.float64).pin_memory()     # callers in a loop pass their buffer back in
    host.copy_(evec, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return host, ev, (evec, evals, ws)         # the device buffers stay referenced until the copy has run


def get_salient_areas(second_smallest_vec):
    avg = np.sum(second_smallest_vec) / len(second_smallest_vec)
    return second_smallest_vec > avg


def neighbour_sets(unique_segments, seg_connectivity):
    """{segment id: set of the ids it points to} for the directed `seg_connectivity` pairs — geometry of the scene, the
    same in every iteration of the cut loop (the reference rebuilds it per call, :181-190: S boolean masks over E pairs,
    6 of the 7 ms a call took on the 625-segment scene)."""
    uniq = np.asarray(unique_segments.cpu() if isinstance(unique_segments, torch.Tensor) else unique_segments)
    conn = np.asarray(seg_connectivity.cpu() if isinstance(seg_connectivity, torch.Tensor) else seg_connectivity)
    out = {int(s): set() for s in uniq}
    if conn.size:
        order = np.argsort(conn[:, 0], kind="stable")
        src, dst = conn[order, 0], conn[order, 1]
        cuts = np.nonzero(np.diff(src))[0] + 1
        for s0, chunk in zip(src[np.concatenate([[0], cuts])].tolist(), np.split(dst, cuts)):
            if s0 in out:
                out[s0] = set(chunk.tolist())
    return out


def separate_segments(bipartition, second_smallest_vec, unique_segments, seg_connectivity, mode="max", neighbours=None):
    """Connected blobs of the foreground side under the directed `seg_connectivity` pairs, merged in the
    reference's scan order (:181-250); returns the blob selected by `mode` as a set of segment ids.
    `neighbours`: neighbour_sets(unique_segments, seg_connectivity), when the caller keeps it across iterations."""
    uniq = np.asarray(unique_segments.cpu() if isinstance(unique_segments, torch.Tensor) else unique_segments)
    if neighbours is None:
        neighbours = neighbour_sets(uniq, seg_connectivity)
    fg_ids = uniq[bipartition]
    # The reference scans ALL blobs for every foreground id (:196-216): the first blob its neighbour set touches takes
    # the id, later touching blobs are merged into that one — with the scan's quirk that the blob right after a merged
    # (popped) one is skipped.  `owner` (id -> its blob) finds the touching blobs directly; only an id that touches two
    # or more blobs needs the reference's positional scan.  Same partition, same blob order.
    blobs, owner = [], {}
    for c in fg_ids.tolist():
        nb = neighbours[c]
        touching = {id(owner[x]) for x in nb if x in owner}
        if not touching:
            blob = {c}
            blobs.append(blob)
            owner[c] = blob
            continue
        if len(touching) == 1:
            blob = owner[next(x for x in nb if x in owner)]
            blob.add(c)
            owner[c] = blob
            continue
        last, k = -1, 0
        while k < len(blobs):
            if not nb.isdisjoint(blobs[k]):
                blobs[k].add(c)
                owner[c] = blobs[k]
                if last != -1:
                    blobs[last] |= blobs[k]
                    for x in blobs[k]:
                        owner[x] = blobs[last]
                    blobs.pop(k)
                else:
                    last = k
            k += 1
    if mode == "max":
        seed_id = int(uniq[int(np.argmax(second_smallest_vec))])
        return next(b for b in blobs if seed_id in b)
    if mode == "avg":
        means = [np.mean(second_smallest_vec[np.isin(uniq, list(b))]) for b in blobs]
        return blobs[int(np.argmax(means))]
    if mode == "largest":
        return blobs[int(np.argmax([len(b) for b in blobs]))]
    if mode == "all":
        return set(int(c) for c in fg_ids)
    raise NotImplementedError(mode)


def segment_ids_to_mask(selected_ids, unique_segments):
    uniq = np.asarray(unique_segments.cpu() if isinstance(unique_segments, torch.Tensor) else unique_segments)
    return np.isin(uniq, list(selected_ids))


def aggregate_features(encoded_features, segment_ids, seg_connectivity, aggregation_mode="mean"):
    """Per-segment mean of the non-zero feature rows, zero segments filled from connected segments
    (reference :350-402, incl. its use of `zero_segments[0]`'s neighbours for every zero segment)."""
    if aggregation_mode not in ("mean", "max"):
        raise ValueError(f"aggregation_mode {aggregation_mode!r}: 'mean' or 'max' (reference :366)")
    dev = encoded_features.device
    unique_segments, inv = torch.unique(segment_ids, return_inverse=True)
    S = unique_segments.shape[0]
    csr = ops.segment_csr(inv.to(torch.int64).contiguous(), S)
    feats = encoded_features.float().contiguous()
    out = torch.empty((S, feats.shape[1]), dtype=torch.float32, device=dev)
    fn = lib.usc_segment_max_nonzero if aggregation_mode == "max" else lib.usc_segment_mean_nonzero
    check(fn(feats.data_ptr(), feats.shape[1], csr.order.data_ptr(), csr.seg_off.data_ptr(), S, out.data_ptr(), None,
             ops._stream()), "usc_segment_mean_nonzero")
    agg = out.clone()
    zero = torch.nonzero(torch.all(agg == 0, dim=-1)).reshape(-1)
    if zero.numel():
        uniq_c, conn = unique_segments.cpu(), seg_connectivity.cpu()
        first = uniq_c[zero[0].item()]
        nbr_ids = conn[conn[:, 0] == first][:, 1]
        nbr_idx = torch.as_tensor([int((uniq_c == s).nonzero(as_tuple=True)[0]) for s in nbr_ids], dtype=torch.long)
        for z in zero.tolist():
            cand = agg[nbr_idx.to(dev)] if nbr_idx.numel() else agg[:0]
            cand = cand[torch.any(cand != 0.0, dim=-1)]
            agg[z] = cand.mean(0) if len(cand) else agg.mean(0)
    return agg, unique_segments


def unscene3d_steps(aggregated_features, unique_segments, seg_connectivity, segment_ids=None, scene_coords=None,
                    scene_colors=None, affinity_tau=0.65, max_number_of_instances=20, similarity_metric="cos",
                    max_extent_ratio=0.8, max_surface_ratio=0.3, eps=1e-5, min_segment_size=4, separation_mode="max",
                    eigvec_hook=None):
    """The masked-NCut loop of `unscene3d` as a generator: it queues one iteration's device work (affinity, degree,
    generalized eigenvector, copy to pinned memory) on the CURRENT stream, yields the event that marks the eigenvector's
    arrival on the host, and — resumed after that event — does the iteration's host logic and queues the next one.
    `StopIteration.value` is the bool[K, S] result.  One host thread can keep several scenes in flight, each on its own
    stream (pseudo_masks/driver.py), without the scenes' host logic fighting over the interpreter lock."""
    num_segments = len(unique_segments)
    if num_segments < 3:
        return np.ones(num_segments, dtype=bool).reshape(1, -1)
    feats = aggregated_features
    dev = (feats[0] if isinstance(feats, tuple) else feats).device
    neighbours = neighbour_sets(unique_segments, seg_connectivity)
    bipartitions, foreground = [], set()
    # The painting (segments of every part cut so far, reference :122-135, :426-427) is host state: S flags, uploaded
    # once per iteration through a pinned buffer.  The reference's `feats = (1 - painting) * feats` per iteration is the
    # ORIGINAL features with the painted rows zeroed (painting only grows): usc_ncut_similarity_masked forms the product
    # while it normalises the rows.  (Round 3 kept painting / mask on the device: five element-wise launches and a
    # blocking pageable upload of the part mask per iteration — 0.9 ms of the 1.8 ms of host time an iteration took.)
    painting_np = np.zeros(num_segments, dtype=bool)
    paint_pin = torch.zeros(num_segments, dtype=torch.uint8).pin_memory()
    paint_dev = torch.zeros(num_segments, dtype=torch.uint8, device=dev)
    host = None
    for it in range(max_number_of_instances):
        if it > 0:
            # safe to rewrite: the upload of the previous iteration ran before the event waited for below fired
            paint_pin.numpy()[:] = painting_np
            paint_dev.copy_(paint_pin, non_blocking=True)
        A, D = get_affinity_matrix(feats, tau=affinity_tau, eps=eps, normalize_sim=True,
                                   similarity_metric=similarity_metric, painting=paint_dev,
                                   zero_rows=paint_dev if it > 0 else None)
        host, event, keep = second_smallest_eigenvector_async(A, D, eps, host=host)
        yield event
        event.synchronize()
        vec = host.numpy().copy()
        del keep
        if eigvec_hook is not None:
            vec = eigvec_hook(it, vec)
        bipartition = get_salient_areas(vec)
        if bipartition.sum() / len(bipartition) > max_extent_ratio:
            bipartition = np.logical_not(bipartition)
            vec = vec * -1
        part = separate_segments(bipartition, vec, unique_segments, seg_connectivity, mode=separation_mode,
                                 neighbours=neighbours)
        painting_np |= np.asarray(segment_ids_to_mask(part, unique_segments), dtype=bool)
        if len(part & foreground) / len(part) > 0.5:
            continue
        if len(part) < min_segment_size:
            continue
        bipartitions.append(segment_ids_to_mask(part - foreground, unique_segments))
        foreground |= part
    return np.stack(bipartitions) if bipartitions else np.zeros((0, num_segments), dtype=bool)


def unscene3d(aggregated_features, unique_segments, seg_connectivity, segment_ids=None, scene_coords=None,
              scene_colors=None, affinity_tau=0.65, max_number_of_instances=20, similarity_metric="cos",
              max_extent_ratio=0.8, max_surface_ratio=0.3, eps=1e-5, min_segment_size=4, separation_mode="max",
              eigvec_hook=None):
    """-> bool[K, S] masks over segments (reference :405-502).

    `eigvec_hook(iteration, vec) -> vec` lets a caller post-process the eigenvector of an iteration.  The
    parity tests use it to impose the sign LAPACK happened to return in the reference run: that sign is
    rounding noise whenever painted (isolated) segments exist and is not reproducible even between two
    scipy installations, yet it steers balanced cuts (0.2 <= foreground ratio <= 0.8)."""
    gen = unscene3d_steps(aggregated_features, unique_segments, seg_connectivity, segment_ids, scene_coords,
                          scene_colors, affinity_tau, max_number_of_instances, similarity_metric, max_extent_ratio,
                          max_surface_ratio, eps, min_segment_size, separation_mode, eigvec_hook)
    try:
        while True:
            next(gen).synchronize()
    except StopIteration as done:
        return done.value
