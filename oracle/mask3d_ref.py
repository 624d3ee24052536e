"""oracle/mask3d_ref.py — CPU (torch) restatement of Mask3D.forward and of the self-training step
(reference models/mask3d.py:200-446, trainer/trainer.py:99-163).  TEST INFRASTRUCTURE ONLY.

Consumes the device model's state_dict.  The decoder blocks are torch.nn.functional primitives
(`F.multi_head_attention_forward` = nn.MultiheadAttention.forward, `F.layer_norm`, matrix products) applied to the
state_dict's tensors, whose reference usage (mask3d.py:491-651) is pinned by tests/golden/decoder_layers.npz; FPS follows sampling_gpu.cu:73-176; the Fourier
encoding follows models/position_embedding.py:12-40,128-157 (pinned by tests/golden/posenc.npz);
the sparse backbone is oracle/res16unet_ref.py (ME semantics — parity unpinned, see sparse_ref.py).
Random key sub-sampling is injected through `randperm(n)` so that device and oracle use the same
indices.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import res16unet_ref as RU
from . import sparse_ref as R


def fps_ref(xyz: np.ndarray, m: int) -> np.ndarray:
    """numpy restatement of furthest_point_sampling_kernel (sampling_gpu.cu:73-176) including the
    block-size dependent tie-break (cuda_utils.h:17-21) and the |p|^2 <= 1e-3 skip."""
    xyz = np.asarray(xyz, np.float32)
    n = xyz.shape[0]
    bs = 1
    while bs * 2 <= n and bs < 512:
        bs *= 2
    tmp = np.full(n, 1e10, np.float32)
    idx = np.zeros(m, np.int32)
    mag = (xyz[:, 0] * xyz[:, 0] + xyz[:, 1] * xyz[:, 1] + xyz[:, 2] * xyz[:, 2]).astype(np.float32)
    ok = ~(mag <= np.float32(1e-3))
    ks = np.arange(n)
    old = 0
    for j in range(1, m):
        d = ((xyz - xyz[old]) ** 2).astype(np.float32)
        d = (d[:, 0] + d[:, 1] + d[:, 2]).astype(np.float32)
        d2 = np.minimum(d, tmp)
        tmp = np.where(ok, d2, tmp)
        cand = np.where(ok, d2, -np.inf)
        best = cand.max()
        if not np.isfinite(best):
            old = 0
        else:
            tied = ks[cand == best]
            old = int(tied[np.lexsort((tied, tied % bs))][0])
        idx[j] = old
    return idx


def fourier_rows(xyz: torch.Tensor, lo, hi, gauss_B: torch.Tensor) -> torch.Tensor:
    xn = ((xyz - lo) * 1.0) / (hi - lo) + 0.0
    proj = (xn * (2 * np.pi)) @ gauss_B
    return torch.cat([proj.sin(), proj.cos()], dim=1)


def _mha(sd, pfx, d, heads, query, key, value, attn_mask=None):
    """nn.MultiheadAttention.forward (sequence-first, dropout 0, need_weights=True) called functionally on the
    state_dict's tensors, so that autograd reaches them (reference models/mask3d.py:491-605 builds the module; its use
    is pinned by tests/golden/decoder_layers.npz and decoder_pass.npz)."""
    return F.multi_head_attention_forward(
        query, key, value, d, heads, sd[pfx + "in_proj_weight"], sd[pfx + "in_proj_bias"], None, None, False, 0.0,
        sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"], training=True, key_padding_mask=None,
        need_weights=True, attn_mask=attn_mask)[0]


def _ln(sd, pfx, d, x):
    return F.layer_norm(x, (d,), sd[pfx + "weight"], sd[pfx + "bias"])


def mask3d_forward(sd: dict, cfg, coords4: np.ndarray, feats: torch.Tensor, raw_xyz: torch.Tensor, point2segment,
                   randperm, dtype=torch.float32, keep_graph=False, attn_hook=None, is_eval=False):
    """-> dict(pred_logits, pred_masks, aux_outputs) like the reference; sd = device model state_dict.
    keep_graph: `sd` already holds CPU tensors of `dtype` (leaves with requires_grad): they are used as they are, so
    that a backward pass from the criterion fills their .grad (full-step gradient / trajectory parity).
    attn_hook(pass_index, mask bool[B,K,Q]) -> mask: lets a test look at / replace the thresholded attention mask of a
    decoder pass — a DISCRETE decision (sigmoid(mean logit) < 0.5, mask3d.py:432-436) that two fp32 evaluation orders
    may take differently for logits within rounding of zero.
    is_eval: the validation / export forward (trainer/trainer.py:384-396 calls `forward(..., is_eval=True)` on the
    module in eval()): batch norms use their running statistics and NO key sub-sampling happens — every voxel of a
    level is a cross-attention key, K = the largest scene's level size (mask3d.py:297-312: `if not (self.max_sample_size
    or is_eval)` guards the truncation), shorter scenes padded with row 0 and masked."""
    m = cfg.model
    d, Q, H = m.hidden_dim, m.num_queries, m.num_heads
    if not keep_graph:
        sd = {k: (v.detach().cpu().to(dtype) if v.dtype.is_floating_point else v.detach().cpu()) for k, v in sd.items()}
    bsd = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
    pyr = RU.Pyramid(coords4)
    layers = (2, 3, 4, 6, 2, 2, 2, 2)
    pcd, aux = RU.res16unet_forward(bsd, pyr, feats.to(dtype), layers, is_eval=is_eval)   # aux = [s16, s8, s4, s2, s1]

    batch_of = [torch.as_tensor(c[:, 0].astype(np.int64)) for c in pyr.coords]     # per level
    nb = int(batch_of[0].max()) + 1
    rows = [[torch.nonzero(b == i).reshape(-1) for i in range(nb)] for b in batch_of]

    # raw-coordinate pyramid by average pooling (no grad)
    with torch.no_grad():
        lvl_xyz = [raw_xyz.to(dtype)]
        for l in range(4):
            lvl_xyz.append(R.avgpool_down2(lvl_xyz[-1], pyr.nbr2[l]))
    gB = sd["pos_enc.gauss_B"]
    pos_enc = []
    for lvl in range(5):                                                           # lvl 0 = s1
        pos_enc.append([fourier_rows(lvl_xyz[lvl][r], lvl_xyz[lvl][r].min(0)[0], lvl_xyz[lvl][r].max(0)[0], gB)
                        for r in rows[lvl]])

    mask_features = pcd @ sd["mask_features_head.kernel"] + sd["mask_features_head.bias"]
    mask_segments = [R.scatter_mean(mask_features[rows[0][b]], point2segment[b]) for b in range(nb)]

    # queries
    c_int = torch.as_tensor(pyr.coords[0][:, 1:].astype(np.float32))
    fps = [torch.as_tensor(fps_ref(c_int[rows[0][b]].numpy(), Q).astype(np.int64)) for b in range(nb)]
    sampled = torch.stack([lvl_xyz[0][rows[0][b]][fps[b]] for b in range(nb)])
    mins = torch.stack([lvl_xyz[0][rows[0][b]].min(0)[0] for b in range(nb)])
    maxs = torch.stack([lvl_xyz[0][rows[0][b]].max(0)[0] for b in range(nb)])
    qpos = torch.stack([fourier_rows(sampled[b], mins[b], maxs[b], gB).T for b in range(nb)])   # B, d, Q
    w0, b0 = sd["query_projection.layers.0.weight"], sd["query_projection.layers.0.bias"]
    w2, b2 = sd["query_projection.layers.2.weight"], sd["query_projection.layers.2.bias"]
    qpos = torch.relu(F.conv1d(torch.relu(F.conv1d(qpos, w0, b0)), w2, b2))
    queries = torch.zeros_like(qpos).permute(0, 2, 1)
    query_pos = qpos.permute(2, 0, 1)

    def mask_module(q, pool_steps, want_attn):
        q = F.layer_norm(q, (d,), sd["decoder_norm.weight"], sd["decoder_norm.bias"])
        me = torch.relu(q @ sd["mask_embed_head.0.weight"].T + sd["mask_embed_head.0.bias"])
        me = me @ sd["mask_embed_head.2.weight"].T + sd["mask_embed_head.2.bias"]
        cls = q @ sd["class_embed_head.weight"].T + sd["class_embed_head.bias"]
        segs = [mask_segments[b] @ me[b].T for b in range(nb)]
        if not want_attn:
            return cls, segs, None
        with torch.no_grad():
            am = torch.zeros(pyr.coords[0].shape[0], Q, dtype=dtype)
            for b in range(nb):
                am[rows[0][b]] = segs[b].detach()[point2segment[b]]
            for l in range(pool_steps):
                am = R.avgpool_down2(am, pyr.nbr2[l])
            am = am.sigmoid() < 0.5
        return cls, segs, am

    pred_cls, pred_masks = [], []
    for dec in range(m.num_decoders):
        for i, hlevel in enumerate(m.hlevels):
            lvl = 4 - hlevel                                                       # aux[hlevel] lives at level lvl
            cls, segs, am = mask_module(queries, lvl, True)
            feats_l = aux[hlevel]
            sizes = [len(r) for r in rows[lvl]]
            if min(sizes) == 1:
                raise RuntimeError("only a single point gives nans in cross-attention")
            K = max(sizes)
            if not (getattr(m, "max_sample_size", False) or is_eval):                 # reference :311-312
                K = min(K, m.sample_sizes[hlevel])
            ridx, midx = [], []
            for b, n in enumerate(sizes):
                if n <= K:
                    idx = torch.zeros(K, dtype=torch.long)
                    idx[:n] = torch.arange(n)
                    mk = torch.ones(K, dtype=torch.bool)
                    mk[:n] = False
                else:
                    idx = randperm(n)[:K]
                    mk = torch.zeros(K, dtype=torch.bool)
                ridx.append(idx)
                midx.append(mk)
            b_aux = torch.stack([feats_l[rows[lvl][b]][ridx[b]] for b in range(nb)])
            b_attn = torch.stack([am[rows[lvl][b]][ridx[b]] for b in range(nb)])
            b_pos = torch.stack([pos_enc[lvl][b][ridx[b]] for b in range(nb)])
            b_attn.permute(0, 2, 1)[b_attn.sum(1) == K] = False
            b_attn = torch.logical_or(b_attn, torch.stack(midx)[..., None])
            if attn_hook is not None:
                b_attn = attn_hook(len(pred_cls), b_attn)
            pfx = f"lin_squeeze.0.{i}."
            src = b_aux.permute(1, 0, 2) @ sd[pfx + "weight"].T + sd[pfx + "bias"]
            ca, sa, ff = f"cross_attention.0.{i}.", f"self_attention.0.{i}.", f"ffn_attention.0.{i}."
            tgt = queries.permute(1, 0, 2)
            upd = _mha(sd, ca + "multihead_attn.", d, H, tgt + query_pos, src + b_pos.permute(1, 0, 2), src,
                       attn_mask=b_attn.repeat_interleave(H, dim=0).permute(0, 2, 1))
            tgt = _ln(sd, ca + "norm.", d, tgt + upd)
            qk = tgt + query_pos
            tgt = _ln(sd, sa + "norm.", d, tgt + _mha(sd, sa + "self_attn.", d, H, qk, qk, tgt))
            hid = torch.relu(tgt @ sd[ff + "linear1.weight"].T + sd[ff + "linear1.bias"])
            tgt = _ln(sd, ff + "norm.", d, tgt + hid @ sd[ff + "linear2.weight"].T + sd[ff + "linear2.bias"])
            queries = tgt.permute(1, 0, 2)
            pred_cls.append(cls)
            pred_masks.append(segs)
    cls, segs, _ = mask_module(queries, 0, False)
    pred_cls.append(cls)
    pred_masks.append(segs)
    return {"pred_logits": pred_cls[-1], "pred_masks": pred_masks[-1],
            "aux_outputs": [{"pred_logits": a, "pred_masks": b} for a, b in zip(pred_cls[:-1], pred_masks[:-1])],
            "backbone_features": pcd, "backbone_levels": aux}
