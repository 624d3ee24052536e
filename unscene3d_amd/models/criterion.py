"""SetCriterion (reference models/criterion.py:94-292): Hungarian matching per prediction level,
weighted cross-entropy over C+1 classes (`eos_coef` on the no-object class, ignore_index 253),
sigmoid-BCE + dice on the matched masks, each divided by the number of target masks.

Differences from the reference are organisational only: all 13 levels' cost matrices are built on
the device first and copied to the host in ONE transfer (the reference syncs 13*B times), then
scipy solves them; the loss arithmetic and its reduction order follow the reference line by line."""
import os

import torch
import torch.nn.functional as F
from torch import nn

from .misc import get_world_size, is_dist_avail_and_initialized

# diagnostic switch (tools/ab.sh): USC3D_NUM_MASKS_ALLREDUCE=0 leaves the reference's num_masks collective out of the device
# criterion — its result is unused there — to price what ONE small collective per step costs next to the step's streams
_NUM_MASKS_ALLREDUCE = __import__("os").environ.get("USC3D_NUM_MASKS_ALLREDUCE", "1") == "1"


def dice_loss(inputs, targets, num_masks: float, weights):
    p = inputs.sigmoid().flatten(1)
    num = 2 * (p * targets).sum(-1)
    den = p.sum(-1) + targets.sum(-1)
    return (weights * (1 - (num + 1) / (den + 1))).sum() / num_masks


def sigmoid_ce_loss(inputs, targets, num_masks: float, weights):
    loss = weights.view(-1, 1) * F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    return loss.mean(1).sum() / num_masks


dice_loss_jit = dice_loss
sigmoid_ce_loss_jit = sigmoid_ce_loss


FUSED = os.environ.get("USC3D_FUSED_CRITERION", "1") == "1"


class _FusedCriterion(torch.autograd.Function):
    """Matching + losses of all prediction levels on the device (csrc/criterion.hip): per scene the cost matrices of
    the 13 levels (2 launches), their assignments (usc_lsap_batch, scipy's algorithm and tie-breaking), the label /
    mask / dice losses (1 launch) — no device->host copy, no host solve — and one table launch for the batch.
    Inputs: class logits f32[L,B,Q,C] and, per scene and level, the mask logits f32[S_b, ld] (ld >= Q: the padded
    tables of models.mask3d._mask_logits are taken as they are).  Output: the [L*4] loss table
    (loss_ce, loss_mask, loss_dice, loss_noise_robust = 0 per level)."""

    @staticmethod
    def forward(ctx, crit, targets, mask_type, logits, *mask_tables):
        import ctypes as C
        from .. import ops
        from .._lib import check, lib
        L, B, Q, NC = logits.shape
        dev = logits.device
        logits = logits.contiguous()
        st = ops._stream()
        m = crit.matcher
        parts = torch.empty((B, L, 4), dtype=torch.float32, device=dev)
        scenes = []
        for b in range(B):
            tabs = [t.contiguous() for t in mask_tables[b * L:(b + 1) * L]]
            S, ld = tabs[0].shape
            tm = targets[b][mask_type]
            labels = targets[b]["labels"].to(torch.int64).contiguous()
            T = int(tm.shape[0])
            tm8 = tm.contiguous().view(torch.uint8) if tm.dtype == torch.bool else (tm != 0).contiguous().view(torch.uint8)
            bits = torch.empty(S, dtype=torch.int32, device=dev)
            cnt = torch.empty(T, dtype=torch.int32, device=dev)
            check(lib.usc_criterion_target_bits(tm8.data_ptr(), T, S, bits.data_ptr(), cnt.data_ptr(), st),
                  "usc_criterion_target_bits")
            ptrs = (C.c_void_p * L)(*[t.data_ptr() for t in tabs])
            cost = torch.empty((L, Q, T), dtype=torch.float32, device=dev)
            comps = torch.empty((3, L, Q, T), dtype=torch.float32, device=dev)      # cmask | cdice | nmat
            ssum = torch.empty((L, Q), dtype=torch.float32, device=dev)
            logp = torch.empty((L, Q, NC), dtype=torch.float32, device=dev)
            wsb = lib.usc_criterion_ws_bytes(L, S, T)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            lg = logits[:, b]                                                       # [L,Q,C] view: strides (B*Q*C, C, 1)
            check(lib.usc_criterion_costs(ptrs, L, ld, S, Q, T, bits.data_ptr(), cnt.data_ptr(), lg.data_ptr(),
                                          B * Q * NC, NC, NC, labels.data_ptr(), float(m.cost_mask), float(m.cost_class),
                                          float(m.cost_dice), cost.data_ptr(), comps[0].data_ptr(), comps[1].data_ptr(),
                                          comps[2].data_ptr(), ssum.data_ptr(), logp.data_ptr(), ws.data_ptr(), wsb, st),
                  "usc_criterion_costs")
            src, tid, status = ops.lsap_batch(cost)                                 # [L,T] queries (ascending), targets
            tcls = torch.empty((L, Q), dtype=torch.int32, device=dev)
            check(lib.usc_criterion_losses(comps[0].data_ptr(), comps[1].data_ptr(), logp.data_ptr(), src.data_ptr(),
                                           tid.data_ptr(), labels.data_ptr(), crit.empty_weight.data_ptr(), L, Q, T, NC,
                                           crit.num_classes, tcls.data_ptr(), parts[b].data_ptr(), st),
                  "usc_criterion_losses")
            scenes.append(dict(tabs=tabs, S=S, ld=ld, T=T, bits=bits, cnt=cnt, src=src, tid=tid, comps=comps, ssum=ssum,
                               logp=logp, tcls=tcls, status=status))
        table = torch.empty((L, 4), dtype=torch.float32, device=dev)
        den_tot = torch.empty(L, dtype=torch.float32, device=dev)
        check(lib.usc_criterion_table(parts.data_ptr(), B, L, table.data_ptr(), den_tot.data_ptr(), st),
              "usc_criterion_table")
        # the differentiable inputs go through save_for_backward (version check: a table modified in place between
        # forward and backward raises instead of differentiating stale numbers); the per-scene intermediates are this
        # Function's own tensors
        ctx.save_for_backward(*[t for sc in scenes for t in sc.pop("tabs")])
        ctx.scenes, ctx.den_tot, ctx.shape, ctx.class_w = scenes, den_tot, (L, B, Q, NC), crit.empty_weight
        crit.last_indices = [[(sc["src"][l], sc["tid"][l]) for sc in scenes] for l in range(L)]   # device tensors
        crit.last_lsap_status = [sc["status"] for sc in scenes]
        crit._queue_status_check(crit.last_lsap_status)
        return table.reshape(-1)

    @staticmethod
    def backward(ctx, dflat):
        import ctypes as C
        from .. import ops
        from .._lib import check, lib
        L, B, Q, NC = ctx.shape
        g = dflat.contiguous()
        dev = g.device
        st = ops._stream()
        dlogits = torch.empty((L, B, Q, NC), dtype=torch.float32, device=dev)
        grads = []
        saved = ctx.saved_tensors
        for b, sc in enumerate(ctx.scenes):
            dtab = torch.empty((L, sc["S"], sc["ld"]), dtype=torch.float32, device=dev)
            ptrs = (C.c_void_p * L)(*[t.data_ptr() for t in saved[b * L:(b + 1) * L]])
            dptrs = (C.c_void_p * L)(*[dtab[l].data_ptr() for l in range(L)])
            check(lib.usc_criterion_backward(ptrs, dptrs, L, sc["ld"], sc["S"], Q, sc["T"], sc["bits"].data_ptr(),
                                             sc["cnt"].data_ptr(), sc["src"].data_ptr(), sc["tid"].data_ptr(),
                                             sc["comps"][2].data_ptr(), sc["ssum"].data_ptr(), sc["logp"].data_ptr(),
                                             sc["tcls"].data_ptr(), ctx.class_w.data_ptr(), g.data_ptr(),
                                             ctx.den_tot.data_ptr(), NC, B * Q * NC, NC, dlogits[:, b].data_ptr(), st),
                  "usc_criterion_backward")
            grads.extend(dtab.unbind(0))
        return (None, None, None, dlogits, *grads)


class LossDict(dict):
    """The criterion's {name: scalar} result plus `flat`: the same scalars as one vector in key order."""
    flat = None


class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses, num_points, oversample_ratio,
                 importance_sample_ratio, class_weights, directions="xyz", use_droploss=False,
                 droploss_iou_thresh=0.1):
        super().__init__()
        self.num_classes = num_classes - 1
        self.class_weights, self.matcher, self.weight_dict = class_weights, matcher, weight_dict
        self.eos_coef, self.losses = eos_coef, list(losses)
        self.use_droploss, self.droploss_iou_thresh = use_droploss, droploss_iou_thresh
        empty_weight = torch.ones(self.num_classes + 1)
        empty_weight[-1] = self.eos_coef
        if self.class_weights != -1:
            assert len(self.class_weights) == self.num_classes, "CLASS WEIGHTS DO NOT MATCH"
            empty_weight[:-1] = torch.tensor(self.class_weights)
        self.register_buffer("empty_weight", empty_weight)
        self.num_points, self.oversample_ratio = num_points, oversample_ratio
        self.importance_sample_ratio = importance_sample_ratio
        self.directions = directions
        self.noise_robust_projection_loss = None    # built lazily: only when loss_noise_robust != 0

    # -- assignment status of the device criterion -------------------------------------------
    # The reference stops a diverged run: scipy's linear_sum_assignment raises ValueError on a cost matrix with NaN /
    # inf entries (matcher.py:163).  The device solver reports that case as status != 0 and the step carries on with an
    # identity assignment, so the status words travel to pinned host memory behind the step's kernels (no
    # synchronisation, like the reducer's late-write flag) and are looked at when the copy has landed: at the next
    # criterion call, or in `check_lsap_status(wait=True)` (before a checkpoint / at the end of a run).
    _STATUS_RING = 8

    def _queue_status_check(self, status_tensors):
        st = [s for s in status_tensors if s is not None and s.is_cuda]
        if not st:
            return
        ring = self.__dict__.setdefault("_status_ring", [])
        self.__dict__["_status_step"] = step = self.__dict__.get("_status_step", 0) + 1
        n = sum(int(s.numel()) for s in st)
        slot = None
        for e in ring:
            if e["free"] and e["host"].numel() >= n:
                slot = e
                break
        if slot is None:
            if len(ring) >= self._STATUS_RING:           # the host is that far ahead: wait for the oldest copy
                self.check_lsap_status(wait=True)
                return self._queue_status_check(status_tensors)
            slot = {"host": torch.empty(max(n, 64), dtype=torch.int32, pin_memory=True), "event": torch.cuda.Event(),
                    "free": True}
            ring.append(slot)
        off = 0
        for s in st:
            slot["host"][off:off + s.numel()].copy_(s.reshape(-1), non_blocking=True)
            off += s.numel()
        slot.update(free=False, n=n, step=step)
        slot["event"].record()

    def check_lsap_status(self, wait=False):
        """Raise ValueError (as scipy does inside the reference's matcher) if an assignment problem of an earlier call
        was infeasible — NaN / inf costs from diverged logits, or a target label outside [0, num_classes] ∪ {253}.
        wait=False looks only at copies that have already landed; wait=True blocks for all pending ones."""
        bad = None
        for e in self.__dict__.get("_status_ring", []):
            if e["free"]:
                continue
            if wait:
                e["event"].synchronize()
            elif not e["event"].query():
                continue
            e["free"] = True
            if bool((e["host"][:e["n"]] != 0).any()) and bad is None:
                bad = e["step"]
        if bad is not None:
            raise ValueError(f"matrix contains invalid numeric entries (assignment costs of criterion call {bad} of this "
                             f"run were NaN / infinite, or a target label was out of range; the call before the current "
                             f"one is call {self.__dict__.get('_status_step', 0)})")

    # -- individual losses ------------------------------------------------------------------
    def loss_labels(self, outputs, targets, indices, num_masks, mask_type, coords=None):
        src_logits = outputs["pred_logits"].float()
        batch_idx, src_idx = self._get_src_permutation_idx(indices)
        matched = torch.cat([t["labels"][J.to(t["labels"].device)] for t, (_, J) in zip(targets, indices)])
        target_classes = torch.full(src_logits.shape[:2], self.num_classes, dtype=torch.int64,
                                    device=src_logits.device)
        target_classes[batch_idx.to(src_logits.device), src_idx.to(src_logits.device)] = matched.to(src_logits.device)
        loss_ce = F.cross_entropy(src_logits.transpose(1, 2), target_classes, self.empty_weight, ignore_index=253)
        return {"loss_ce": loss_ce}

    def loss_masks(self, outputs, targets, indices, num_masks, mask_type="masks", coords=None):
        l_mask, l_dice, l_noise = [], [], []
        for b, (map_id, target_id) in enumerate(indices):
            pred = outputs["pred_masks"][b]
            dev = pred.device
            m = pred[:, map_id.to(dev)].T                                   # [T, S]
            tgt = targets[b][mask_type][target_id.to(targets[b][mask_type].device)]
            if self.weight_dict["loss_noise_robust"] != 0:
                from .noise_robust_loss import ProjectionMaskLoss
                if self.noise_robust_projection_loss is None:
                    self.noise_robust_projection_loss = ProjectionMaskLoss(directions=self.directions)
                sampled = m[:, targets[b]["point2segment"]] if coords.shape[0] != m.shape[1] else m
                bmask = coords[:, 0] == b
                bl, all_shape = self.noise_robust_projection_loss(
                    sampled, targets[b]["masks"][target_id.to(targets[b]["masks"].device)].float(), coords[bmask])
                l_noise.append(bl / all_shape)
            else:
                l_noise.append(torch.as_tensor(0.0, dtype=torch.float32, device=dev))
            if self.num_points != -1:
                pidx = torch.randperm(tgt.shape[1], device=tgt.device)[:int(self.num_points * tgt.shape[1])]
                m, tgt = m[:, pidx], tgt[:, pidx]
            n_here = tgt.shape[0]
            if self.use_droploss:
                fg = m > 0.0
                iou = (fg * tgt).sum(dim=1) / (fg + tgt).sum(dim=1)
                weights = (iou >= self.droploss_iou_thresh).float()
            else:
                weights = torch.ones(m.shape[0], device=dev)
            tgt = tgt.float()
            l_mask.append(sigmoid_ce_loss(m, tgt, n_here, weights))
            l_dice.append(dice_loss(m, tgt, n_here, weights))
        return {"loss_mask": torch.sum(torch.stack(l_mask)), "loss_dice": torch.sum(torch.stack(l_dice)),
                "loss_noise_robust": torch.sum(torch.stack(l_noise))}

    @staticmethod
    def _get_src_permutation_idx(indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        return batch_idx, torch.cat([src for (src, _) in indices])

    @staticmethod
    def _get_tgt_permutation_idx(indices):
        batch_idx = torch.cat([torch.full_like(tgt, i) for i, (_, tgt) in enumerate(indices)])
        return batch_idx, torch.cat([tgt for (_, tgt) in indices])

    def get_loss(self, loss, outputs, targets, indices, num_masks, mask_type, coords=None):
        table = {"labels": self.loss_labels, "masks": self.loss_masks}
        assert loss in table, f"do you really want to compute {loss} loss?"
        return table[loss](outputs, targets, indices, num_masks, mask_type, coords)

    # -- matching of every level with one host transfer ------------------------------------
    def match_all_levels(self, levels, targets, mask_type):
        """Cost matrices of ALL prediction levels in a handful of batched launches (the reference builds
        them level by level, matcher.py:98-168), one device->host copy, then scipy per (level, scene)."""
        m = self.matcher
        forced = getattr(self, "forced_indices", None)
        if forced is not None:      # parity tests: impose another run's assignment (a discrete decision)
            self._labels_host = [t["labels"].detach().cpu().to(torch.int64) for t in targets]
            return [[(s_.cpu().to(torch.int64), t_.cpu().to(torch.int64)) for s_, t_ in lv] for lv in forced]
        if m.num_points != -1:      # random point sub-sampling: keep the reference's per-level path
            per_level = [m.cost_matrices(lv, targets, mask_type) for lv in levels]
        else:
            per_level = [[None] * len(targets) for _ in levels]
            L = len(levels)
            with torch.no_grad():
                for b, tgt in enumerate(targets):
                    logits = torch.stack([lv["pred_logits"][b] for lv in levels]).float()          # [L,Q,C]
                    masks = torch.stack([lv["pred_masks"][b] for lv in levels]).float().transpose(1, 2)   # [L,Q,S]
                    tgt_ids = tgt["labels"].clone()
                    ignore = tgt_ids == 253
                    tgt_ids[ignore] = 0
                    cost_class = -logits.softmax(-1)[:, :, tgt_ids]                                   # [L,Q,T]
                    cost_class[:, :, ignore] = -1.0
                    tm = tgt[mask_type].to(masks)                                                     # [T,S]
                    S = masks.shape[2]
                    pos = F.binary_cross_entropy_with_logits(masks, torch.ones_like(masks), reduction="none")
                    neg = F.binary_cross_entropy_with_logits(masks, torch.zeros_like(masks), reduction="none")
                    cost_mask = (pos @ tm.T + neg @ (1 - tm).T) / S
                    p = masks.sigmoid()
                    cost_dice = 1 - (2 * (p @ tm.T) + 1) / (p.sum(-1)[:, :, None] + tm.sum(-1)[None, None, :] + 1)
                    C = m.cost_mask * cost_mask + m.cost_class * cost_class + m.cost_dice * cost_dice
                    for l in range(L):
                        per_level[l][b] = C[l]
        flat = [C for Cs in per_level for C in Cs]
        widths = [C.shape[1] for C in flat]
        costs = torch.cat(flat, dim=1)
        # the target labels ride along as one extra row (class ids are exact in fp32), so that the batched losses
        # need no device->host copy of their own
        labels = torch.cat([t["labels"].reshape(-1) for t in targets]).to(costs)
        row = torch.zeros((1, costs.shape[1]), dtype=costs.dtype, device=costs.device)
        row[0, :labels.shape[0]] = labels
        host = torch.cat([costs, row], dim=0).cpu()                           # the single D2H of the step
        lab = host[-1, :labels.shape[0]].to(torch.int64)
        self._labels_host = list(torch.split(lab, [t["labels"].numel() for t in targets]))
        pieces = torch.split(host[:-1], widths, dim=1)
        out, q = [], 0
        for Cs in per_level:
            out.append([m.solve(pieces[q + b]) for b in range(len(Cs))])
            q += len(Cs)
        return out

    def _batched_losses(self, levels, targets, all_indices, mask_type):
        """All levels' classification and mask losses in batched launches; returns the same 4 scalars per
        level as get_loss (arithmetic per level identical to loss_labels / loss_masks)."""
        L, B = len(levels), len(targets)
        dev = levels[0]["pred_logits"].device
        # ---- labels (weighted CE, mean over B*Q with weights, ignore_index 253: F.cross_entropy semantics)
        logits = torch.stack([lv["pred_logits"] for lv in levels]).float()                # [L,B,Q,C]
        Q = logits.shape[2]
        tc = torch.full((L, B, Q), self.num_classes, dtype=torch.int64)
        labels_host = self._labels_host            # filled by match_all_levels from the step's single D2H copy
        for l in range(L):
            for b, (src, J) in enumerate(all_indices[l]):
                tc[l, b, src] = labels_host[b][J]
        tc = tc.to(dev)
        nll = F.cross_entropy(logits.reshape(L * B * Q, -1), tc.reshape(-1), self.empty_weight, ignore_index=253,
                              reduction="none").reshape(L, -1)
        w = self.empty_weight[tc.clamp(max=self.num_classes)].reshape(L, -1) * (tc != 253).reshape(L, -1)
        loss_ce = nll.sum(1) / w.sum(1)                                                    # [L]
        # ---- masks: per scene, all levels at once
        loss_mask = torch.zeros(L, device=dev)
        loss_dice = torch.zeros(L, device=dev)
        for b, tgt in enumerate(targets):
            pm = torch.stack([lv["pred_masks"][b] for lv in levels])                       # [L,S,Q]
            src = torch.stack([all_indices[l][b][0] for l in range(L)]).to(dev)            # [L,T]
            tid = torch.stack([all_indices[l][b][1] for l in range(L)]).to(dev)            # [L,T]
            mp = torch.gather(pm, 2, src[:, None, :].expand(-1, pm.shape[1], -1)).transpose(1, 2)   # [L,T,S]
            tm = tgt[mask_type].to(dev)[tid].float()                                      # [L,T,S]
            T = tm.shape[1]
            if self.use_droploss:
                fg = mp > 0.0
                iou = (fg * tm).sum(2) / (fg + tm).sum(2)
                wts = (iou >= self.droploss_iou_thresh).float()
            else:
                wts = torch.ones(mp.shape[:2], device=dev)
            bce = F.binary_cross_entropy_with_logits(mp, tm, reduction="none")
            loss_mask = loss_mask + (wts[..., None] * bce).mean(2).sum(1) / T
            p = mp.sigmoid()
            dice = 1 - (2 * (p * tm).sum(2) + 1) / (p.sum(2) + tm.sum(2) + 1)
            loss_dice = loss_dice + (wts * dice).sum(1) / T
        # one [L, 4] table; the per-key scalars are views of it.  `flat` hands the table to the trainer, whose weighted
        # sum then differentiates through ONE stack instead of 4 L selects (~120 tiny backward launches per step)
        table = torch.stack([loss_ce, loss_mask, loss_dice, torch.zeros(L, dtype=torch.float32, device=dev)], dim=1)
        flat = table.reshape(-1)
        out = LossDict()
        for l in range(L):
            sfx = "" if l == 0 else f"_{l - 1}"
            for j, name in enumerate(("loss_ce", "loss_mask", "loss_dice", "loss_noise_robust")):
                out[name + sfx] = flat[4 * l + j]
        out.flat = flat
        return out

    def _fused_tables(self, levels, targets, mask_type):
        """The per-(scene, level) mask-logit tables [S, ld] for the device criterion, or None when the fused path does
        not apply (CPU tensors, sub-sampled points, drop loss, noise-robust loss, > 32 targets or > 128 queries ...)."""
        if not (FUSED and self.losses == ["labels", "masks"] and self.num_points == -1 and self.matcher.num_points == -1
                and self.weight_dict.get("loss_noise_robust", 0) == 0 and not self.use_droploss and targets
                and 1 <= len(levels) <= 16):
            return None
        lg = levels[0]["pred_logits"]
        if not (lg.is_cuda and lg.dtype == torch.float32 and lg.dim() == 3 and lg.shape[1] <= 128
                and lg.shape[2] == self.num_classes + 1):
            return None
        Q = lg.shape[1]
        tables = []
        for b, tgt in enumerate(targets):
            tm = tgt.get(mask_type)
            if tm is None or not tm.is_cuda or not (1 <= tm.shape[0] <= min(32, Q)) or "labels" not in tgt:
                return None
            lab = tgt["labels"]
            if not (torch.is_tensor(lab) and lab.is_cuda and lab.numel() == tm.shape[0]):
                return None                                  # host labels / wrong count: the operator path handles them
            ld = None
            for lv in levels:
                t = lv["pred_masks"][b]
                t = getattr(t, "_usc_padded", t)                 # models.mask3d._mask_logits: the padded [S, 128] table
                if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.is_contiguous()
                        and Q <= t.shape[1] <= 128 and t.shape[0] == tm.shape[1] and t.shape[1] == (ld or t.shape[1])):
                    return None
                ld = t.shape[1]
                tables.append(t)
        return tables

    def forward(self, outputs, targets, mask_type, coords=None):
        final = {k: v for k, v in outputs.items() if k != "aux_outputs"}
        levels = [final] + list(outputs.get("aux_outputs", []))
        self.check_lsap_status()                             # an earlier call's infeasible assignment -> ValueError
        tables = self._fused_tables(levels, targets, mask_type)
        if tables is None and FUSED and levels[0]["pred_logits"].is_cuda and not self.__dict__.get("_warned_operator_path"):
            # e.g. a prediction table that lost its `_usc_padded` companion by being cloned / re-wrapped, host-side
            # targets, > 32 targets: correct, but ~150 stock launches and a device->host copy per step slower
            self.__dict__["_warned_operator_path"] = True
            import warnings
            warnings.warn("SetCriterion: the device criterion (csrc/criterion.hip) does not apply to these inputs; "
                          "using the torch-operator path (see SetCriterion._fused_tables for the conditions)")
        if tables is not None:
            if is_dist_avail_and_initialized() and _NUM_MASKS_ALLREDUCE:   # the reference's collective (criterion.py:258-260); its result is
                # never used by the losses (loss_masks overwrites num_masks, :189), so nobody waits for it here
                # (a fill launch, not torch.as_tensor(list, device=...): that is a copy from pageable host memory, which
                #  makes the HOST wait for everything queued on the stream — in the middle of the step.  Measured with a
                #  one-rank RCCL group, bench.py --force-dist: 26.3-26.6 ms per step with it, 23.8 without, round 6)
                nm = torch.full((1,), float(sum(len(t["labels"]) for t in targets)), dtype=torch.float, device=tables[0].device)
                torch.distributed.all_reduce(nm)
            logits = torch.stack([lv["pred_logits"] for lv in levels])                       # [L,B,Q,C]
            flat = _FusedCriterion.apply(self, targets, mask_type, logits, *tables)
            out = LossDict()
            for l in range(len(levels)):
                sfx = "" if l == 0 else f"_{l - 1}"
                for j, name in enumerate(("loss_ce", "loss_mask", "loss_dice", "loss_noise_robust")):
                    out[name + sfx] = flat[4 * l + j]
            out.flat = flat
            return out
        all_indices = self.match_all_levels(levels, targets, mask_type)
        self.last_indices = all_indices

        num_masks = sum(len(t["labels"]) for t in targets)
        if is_dist_avail_and_initialized():     # the reference's only explicit collective (criterion.py:258-260)
            nm = torch.as_tensor([num_masks], dtype=torch.float, device=outputs["pred_logits"].device)
            torch.distributed.all_reduce(nm)
            num_masks = torch.clamp(nm / get_world_size(), min=1).item()
        else:
            num_masks = max(float(num_masks), 1.0)

        batchable = (self.losses == ["labels", "masks"] and self.num_points == -1
                     and self.weight_dict.get("loss_noise_robust", 0) == 0)
        same_T = all(len(all_indices[l][b][0]) == len(all_indices[0][b][0]) for l in range(len(levels))
                     for b in range(len(targets)))
        if batchable and same_T:
            return self._batched_losses(levels, targets, all_indices, mask_type)

        losses = {}
        for loss in self.losses:
            losses.update(self.get_loss(loss, final, targets, all_indices[0], num_masks, mask_type, coords))
        for i, aux in enumerate(levels[1:]):
            for loss in self.losses:
                ld = self.get_loss(loss, aux, targets, all_indices[i + 1], num_masks, mask_type, coords)
                losses.update({f"{k}_{i}": v for k, v in ld.items()})
        return losses
