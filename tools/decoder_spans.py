#!/usr/bin/env python
"""Compute-stream time of the sections of the bench step, free-running (events on the default stream, no host syncs
added): backbone forward | decoder forward | criterion | criterion + decoder backward | backbone backward | optimizer.
Made to see what a second stream (USC3D_KV_SIDE_STREAM) takes off the compute stream, section by section.
Usage (GPU box): python tools/decoder_spans.py [bench.py flags]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

args = bench.parse()
dev = torch.device("cuda:0")
step = bench.make_mask3d_step(args, dev, 0, 1)
model = step.module.model
main = torch.cuda.default_stream(0)
marks = []            # per step: dict name -> event


def mark(name):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(main)
    marks[-1][name] = ev
    if name == "dec_bwd_done":          # where the key-preparation stream is when the host has issued the decoder's backward
        for st in model.__dict__.get("_usc_side_streams", {}).values():
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(st)
            marks[-1]["side_done"] = ev


orig_bb = model.backbone.forward


def bb_forward(*a, **k):
    mark("bb0")
    out = orig_bb(*a, **k)
    mark("bb1")
    feats = [t.F for t in (out[1] if isinstance(out, tuple) else [out]) if hasattr(t, "F")]
    if isinstance(out, tuple) and hasattr(out[0], "F"):
        feats.append(out[0].F)
    for f in feats:
        if f.requires_grad:
            f.register_hook(lambda g: (mark("dec_bwd_done"), None)[1])      # the last firing overwrites: all decoder gradients in
    return out


model.backbone.forward = bb_forward
orig_fwd = model.forward


def fwd(*a, **k):
    out = orig_fwd(*a, **k)
    mark("dec_fwd_done")
    return out


model.forward = fwd
orig_backward = torch.Tensor.backward


def backward(self, *a, **k):
    mark("bwd0")
    # the end-of-backward callbacks (lane join, side-stream join) run inside orig_backward: note where the lane is BEFORE
    # the compute stream waits for it — an engine callback queued first runs first
    from unscene3d_amd import units

    def note():
        for key, ent in units._LANE.items():
            if ent is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(ent[0])
                marks[-1]["lane_done"] = ev
        mark("chain_done")
    torch.autograd.Variable._execution_engine.queue_callback  # (only callable during backward: registered by a hook below)
    self.register_hook(lambda g: (torch.autograd.Variable._execution_engine.queue_callback(note), None)[1])
    r = orig_backward(self, *a, **k)
    mark("bwd1")
    return r


torch.Tensor.backward = backward
n = args.steps + args.warmup
import time  # noqa: E402
htimes = []          # per step: {call name: host ms}


def timed_call(obj, name, label):
    orig = getattr(obj, name)

    def wrapper(*a, **k):
        t = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            htimes[-1][label] = htimes[-1].get(label, 0.0) + (time.perf_counter() - t) * 1e3
    setattr(obj, name, wrapper)


if step.prefetch is not None:              # where does the compute stream spend the time between the step's start and the backbone?
    _take = step.prefetch.take

    def take_marked(*a, **k):
        mark("take0")
        out = _take(*a, **k)
        mark("take1")
        return out
    step.prefetch.take = take_marked
timed_call(step.module, "training_step", "training_step (forward + criterion)")
timed_call(torch.Tensor, "backward", "backward")
timed_call(step.opt, "step", "optimizer")
if step.prefetch is not None:
    timed_call(step.prefetch, "take", "prefetch.take (waits for the worker thread to have issued the batch)")
    timed_call(step.prefetch, "submit", "prefetch.submit")
    timed_call(step.prefetch, "_issue", "worker thread: issuing the next batch (overlaps the main thread)")
host = []
for i in range(n):
    if i == args.warmup:
        torch.cuda.synchronize()      # like bench.py in front of its timed loop: the first step after it starts on an idle device
    marks.append({})
    htimes.append({})
    mark("t0")
    h0 = time.perf_counter()
    step(1)
    host.append((time.perf_counter() - h0) * 1e3)
    mark("t1")
torch.cuda.synchronize()
print(f"host time to issue a step: first after the synchronize {host[args.warmup]:.2f} ms, others {sum(host[args.warmup + 1:]) / max(1, n - args.warmup - 1):.2f} ms")
rows = [("step start -> prefetch.take() entered", "t0", "take0"), ("inside prefetch.take() (wait for the batch's event)", "take0", "take1"),
        ("take() returned -> backbone forward entered", "take1", "bb0"),
        ("backbone forward", "bb0", "bb1"), ("decoder forward", "bb1", "dec_fwd_done"), ("criterion", "dec_fwd_done", "bwd0"),
        ("criterion + decoder backward", "bwd0", "dec_bwd_done"), ("backbone backward", "dec_bwd_done", "bwd1"),
        ("reduce + optimizer", "bwd1", "t1"), ("step", "t0", "t1"),
        ("weight-gradient lane done AFTER the compute stream's last backward kernel by", "chain_done", "lane_done"),
        ("key-preparation stream done AFTER the compute stream's decoder backward by", "dec_bwd_done", "side_done")]
for label in htimes[args.warmup]:
    rest = [h.get(label, 0.0) for h in htimes[args.warmup + 1:]]
    print(f"   host: {label}: first {htimes[args.warmup][label]:.2f} ms, others {sum(rest) / max(1, len(rest)):.2f} ms")
first, use = marks[args.warmup], marks[args.warmup + 1:]
for name, a, b in rows:
    v = [m[a].elapsed_time(m[b]) for m in use if a in m and b in m]
    f = first[a].elapsed_time(first[b]) if a in first and b in first else float("nan")
    print(f"{sum(v) / max(1, len(v)):8.3f} ms  {name}   (first step after the synchronize: {f:.3f})")
step.close()
