"""Training step of the self-training loop (reference trainer/trainer.py:44-163, :953-966).

`InstanceSegmentation` mirrors the LightningModule's constructor logic (model + matcher + criterion
with the 13-level weight_dict), `training_step` and `configure_optimizers`, without depending on
PyTorch-Lightning / hydra (neither is part of the accelerated path).  Differences: losses are summed
on the device and synchronised once per step (the reference does 52 `.cpu().item()` calls, :149)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import MinkowskiEngine as ME
from ..config import instantiate_model
from ..models.criterion import SetCriterion
from ..models.mask3d import SINGLE_POINT_ERROR
from ..models.matcher import HungarianMatcher


def prepare_steady_state(device, main_factor: float = 1.75, side_bytes: int = 768 << 20, min_main_bytes: int = 3 << 30,
                         freeze_heap: bool = True) -> dict:
    """Call ONCE, after the first one or two training steps (every second stream has been picked, the decoder passes are
    captured, the step's high-water mark is known) and before the run's steady state.

    * Memory.  The step allocates from four stream pools of torch's caching allocator (compute stream, weight-gradient
      lane, key-preparation stream, prefetcher) and the scenes differ in size, so every new largest scene grew a pool by
      a `hipMalloc` in the middle of a step: 384 ms steps among 132 ms ones at eight scenes per GPU, 26-37 ms steps
      among 23.7 ms ones at one (round 5: `profiles/r05_soak.json`, `r05_scenes_per_gpu.txt`), and 22 GB reserved for
      5.7 GB in use.  Here the cached segments are handed back once (`empty_cache`, a device synchronise) and every pool
      gets ONE block — `main_factor` x the peak allocation so far on the compute stream, `side_bytes` on each second
      stream — which is freed at once and stays cached: the allocator carves later requests out of it instead of asking
      the driver.  MI355X has 288 GB; a training process should never be inside `hipMalloc` after its first steps.
      Measured together with `StepsInFlight(2)` over 400 steps on eight scenes of 119 k ... 178 k voxels
      (`profiles/r06_inflight_ab.txt`): reserved memory flat at 14.5 GB from the call on (23-24 GB and still growing
      without the bound on the steps in flight), p50 / p99 / max 23.7-24.0 / 26.5 / 27.2-27.6 ms; with factors of
      1.4 / 512 MB the pools grew once more (11.8 -> 11.9 GB, max 29.8 ms); without this call (bound alone) 11.2 GB
      reserved and a 91 ms step while the pools found their size.
    * Interpreter heap.  The module tree, the plans and the captured graphs are ~10^5 container objects that live for
      the whole run; every generation-2 pass of CPython's cycle collector walks them (a multi-millisecond pause on the
      thread that issues the step).  `gc.freeze()` moves what exists now out of the collector's sight; the per-step
      garbage (autograd nodes, lists of tensors) is still collected.

    Reference counterpart: none — the reference trains through PyTorch-Lightning's loop on one stream
    (trainer/trainer.py:99-163) with DataLoader worker PROCESSES (conf/data/indoor.yaml:24-25).
    -> what was done: {"main_bytes", "side_bytes", "streams", "reserved_before", "reserved_after", "frozen_objects"}."""
    import gc

    from .. import streams
    device = torch.device(device)
    out = {}
    if device.type == "cuda":
        idx = device.index if device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(idx):
            torch.cuda.synchronize()
            peak = torch.cuda.max_memory_allocated()
            out["reserved_before"] = torch.cuda.memory_reserved()
            torch.cuda.empty_cache()
            main_bytes = max(int(min_main_bytes), int(main_factor * peak))
            main = torch.cuda.current_stream()
            side = [s for (d, _), s in streams._PICKED.items() if d == idx and s.cuda_stream != main.cuda_stream]
            for st, nbytes in [(main, main_bytes)] + [(s, int(side_bytes)) for s in side]:
                with torch.cuda.stream(st):
                    block = torch.empty(nbytes, dtype=torch.uint8, device=device)
                    del block                    # back to the stream's pool (nothing was launched on it)
            torch.cuda.synchronize()
            out.update(main_bytes=main_bytes, side_bytes=int(side_bytes), streams=1 + len(side),
                       reserved_after=torch.cuda.memory_reserved())
    if freeze_heap:
        gc.collect()
        gc.freeze()
        out["frozen_objects"] = gc.get_freeze_count()
    return out


class StepsInFlight:
    """At most `depth` training steps queued on the device: `begin()` in front of a step waits (interpreter lock released)
    until the step issued `depth` steps ago has finished, `end()` behind the optimizer step marks this one.

    The loop never reads a loss back, so nothing else bounds the host: it issues a 150 k-voxel step in 14-19 ms, the
    device needs 23.7, and every step the host ran 5-10 ms further ahead.  The caching allocator paid for it: a block
    that a SECOND stream has read (`record_stream`: the prefetched batch, the feature maps the key-preparation stream
    samples, what the weight-gradient lane holds) is not reusable before that stream has passed the free, so with the
    host seconds ahead every pool kept several steps' worth of such blocks — 22-24 GB reserved for 5.7 GB in use, grown by
    a `hipMalloc` inside a step whenever the lead reached a new maximum (the 37-104 ms steps of the soak runs,
    `profiles/r05_soak.json`, `r06_soak_ab.txt`).  Two steps in flight keep the device fed — it always has a whole step
    queued — and bound what the pools must hold.  The reference's loop is bounded by construction: 52 `.cpu().item()`
    read-backs per step (trainer/trainer.py:149).  depth <= 0: unbounded (the behaviour up to round 5)."""

    def __init__(self, depth: int = 2):
        import collections
        self.depth = int(depth)
        self._marks = collections.deque()

    def begin(self):
        while self.depth > 0 and len(self._marks) >= self.depth:
            self._marks.popleft().synchronize()

    def end(self, stream=None):
        """-> the step's event (None when unbounded): completes when the device has finished everything queued on the
        compute stream up to here — call it behind the optimizer step, i.e. behind the end-of-backward joins of the
        second streams (`ScenePrefetcher.retire` takes it)."""
        if self.depth > 0:
            ev = torch.cuda.Event()
            ev.record(stream if stream is not None else torch.cuda.current_stream())
            self._marks.append(ev)
            return ev
        return None


class InstanceSegmentation(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.mask_type = "segment_mask" if config.model.train_on_segments else "masks"
        self.model = instantiate_model(config)
        m = config.matcher
        matcher = HungarianMatcher(cost_class=m.cost_class, cost_mask=m.cost_mask, cost_dice=m.cost_dice,
                                   cost_noise_robust=m.cost_noise_robust, num_points=m.num_points)
        weight_dict = {"loss_ce": matcher.cost_class, "loss_mask": matcher.cost_mask, "loss_dice": matcher.cost_dice,
                       "loss_noise_robust": matcher.cost_noise_robust}
        aux = {}
        for i in range(self.model.num_levels * self.model.num_decoders):
            ignored = i in config.general.ignore_mask_idx
            aux.update({f"{k}_{i}": (0.0 if ignored else v) for k, v in weight_dict.items()})
        weight_dict.update(aux)
        l = config.loss
        self.criterion = SetCriterion(num_classes=l.num_classes, matcher=matcher, weight_dict=weight_dict,
                                      eos_coef=l.eos_coef, losses=l.losses, num_points=l.num_points,
                                      oversample_ratio=l.oversample_ratio,
                                      importance_sample_ratio=l.importance_sample_ratio,
                                      class_weights=l.class_weights, directions=l.directions,
                                      use_droploss=l.use_droploss, droploss_iou_thresh=l.droploss_iou_thresh)

    def forward(self, x, point2segment=None, raw_coordinates=None, is_eval=False, num_segments=None):
        return self.model(x, point2segment, raw_coordinates=raw_coordinates, is_eval=is_eval,
                          num_segments=num_segments)

    def training_step(self, batch, batch_idx=0):
        """-> (total weighted loss tensor, dict of weighted loss tensors) or None when the step is skipped."""
        data, target, file_names = batch
        if data.features.shape[0] > self.config.general.max_batch_size:
            raise RuntimeError("BATCH TOO BIG")
        if len(target) == 0:
            return None
        x = getattr(data, "sparse_tensor", None)      # built ahead of time by datasets.prefetch.ScenePrefetcher
        if x is not None:
            raw_coordinates = data.raw_coordinates
        else:
            feats, raw_coordinates = data.features, None
            if self.config.data.add_raw_coordinates:
                raw_coordinates = feats[:, -3:].contiguous()
                feats = feats[:, :-3].contiguous()
            dev = next(self.parameters()).device
            x = ME.SparseTensor(coordinates=data.coordinates, features=feats, device=dev)
        try:
            ns = [t.get("num_segments") for t in target]
            output = self.forward(x, point2segment=[t["point2segment"] for t in target],
                                  raw_coordinates=raw_coordinates,
                                  num_segments=None if any(n is None or n.is_cuda for n in ns) else ns)
        except RuntimeError as err:
            if err.args and err.args[0] == SINGLE_POINT_ERROR:
                return None
            raise
        losses = self.criterion(output, target, mask_type=self.mask_type, coords=x.C)
        wd = self.criterion.weight_dict
        # reference :137-149: `losses[k] *= weight_dict[k]` per key, then sum(losses.values()).  One multiply and one
        # ordered sum over the stacked scalars instead of ~50 scalar multiplies and ~50 scalar adds (and their
        # backward nodes); the per-key weighted losses are views of the product.
        keys = [k for k in losses if k in wd]
        flat = getattr(losses, "flat", None)
        if flat is not None and len(keys) == len(losses) == flat.numel():
            vals = flat                                   # the criterion's own table, already in key order
        else:
            vals = torch.stack([losses[k] for k in keys])
        wkey = (tuple(keys), vals.device)
        if getattr(self, "_wvec_key", None) != wkey:
            self._wvec = torch.tensor([float(wd[k]) for k in keys], dtype=vals.dtype, device=vals.device)
            self._wvec_key = wkey
        wv = vals * self._wvec
        # the reference's python sum() adds in key order; torch.sum uses a tree: equal to fp32 round-off (1e-7)
        return wv.sum(), dict(zip(keys, wv.unbind(0)))

    @torch.no_grad()
    def eval_step(self, batch, batch_idx=0, label_offset=0):
        """Validation / export step (reference trainer/trainer.py:367-443): `forward(..., is_eval=True)` — no key
        sub-sampling (models/mask3d.py:311-312), batch norms on their running statistics when the module is in
        eval() — the weighted validation losses, and the prediction half of `eval_instance_step` (:479-651) on the
        device (trainer/postprocess.py).  Between self-training rounds this runs over every train + val scene to export
        the next round's masks (`general.save_for_freemask`, :743-760).
        -> None (skipped scene), or {"losses": {val_<k>: float}, "instances": [per scene dict]}."""
        from .postprocess import export_instances, save_for_freemask
        data, target, file_names = batch
        # (the reference returns 0. for an empty batch, :373-381; its "no targets" early exit is commented out, :369-371 —
        # an unlabeled export run has targets that only carry point2segment)
        if data.features.shape[0] == 0 or len(target) == 0:
            return None
        # a deferred assignment-status word of an earlier TRAINING step must not surface as this scene's error
        chk = getattr(self.criterion, "check_lsap_status", None)
        if chk is not None:
            chk(wait=True)
        feats, raw_coordinates = data.features, None
        if self.config.data.add_raw_coordinates:
            raw_coordinates = feats[:, -3:].contiguous()
            feats = feats[:, :-3].contiguous()
        dev = next(self.parameters()).device
        x = ME.SparseTensor(coordinates=data.coordinates, features=feats, device=dev)
        try:
            ns = [t.get("num_segments") for t in target]
            output = self.forward(x, point2segment=[t["point2segment"] for t in target],
                                  raw_coordinates=raw_coordinates, is_eval=True,
                                  num_segments=None if any(n is None or n.is_cuda for n in ns) else ns)
        except RuntimeError as err:
            if err.args and err.args[0] == SINGLE_POINT_ERROR:
                return None
            raise
        val = {}
        if getattr(self.config.data, "test_mode", "validation") != "test":     # :400: no criterion on the test split
            losses = self.criterion(output, target, mask_type=self.mask_type, coords=x.C)
            wd = self.criterion.weight_dict
            keys = [k for k in losses if k in wd]
            host = torch.stack([losses[k].detach() for k in keys]).cpu().tolist()       # one read-back, not 52
            val = {f"val_{k}": v * wd[k] for k, v in zip(keys, host)}                   # :410-416, :441
        g = self.config.general
        instances = export_instances(output, target, data.target_full, data.inverse_maps, raw_coordinates, g,
                                     num_classes=self.model.num_classes, decoder_id=g.decoder_id,
                                     train_on_segments=self.model.train_on_segments,
                                     eval_on_segments=g.eval_on_segments, label_offset=label_offset,
                                     full_res_coords=data.full_res_coords)
        if getattr(g, "save_for_freemask", False):
            for name, coords, inst in zip(file_names, data.full_res_coords, instances):
                save_for_freemask(g.save_dir, name, coords, inst["pred_masks"])
        sc = output.get("sampled_coords")
        if sc is not None and not isinstance(sc, np.ndarray):       # the reference hands back a numpy array (mask3d.py:467)
            output["sampled_coords"] = np.asarray(sc)
        return {"losses": val, "instances": instances, "output": output}

    def configure_optimizers(self, steps_per_epoch: int, epochs: int = None):
        o = self.config.optimizer
        optimizer = torch.optim.AdamW(self.parameters(), lr=o.lr)
        scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=o.lr,
                                                        epochs=epochs or self.config.trainer.max_epochs,
                                                        steps_per_epoch=steps_per_epoch)
        return optimizer, scheduler
