"""HIP streams that really run beside each other.

HIP multiplexes its streams onto a few hardware queues — GPU_MAX_HW_QUEUES, 4 by default, per priority — and hands a new
stream the least-used queue.  torch draws its streams from a pool of 32 per priority, so which queue a
`torch.cuda.Stream()` lands on depends on how many streams the process created before; two streams on one queue execute
strictly one kernel after the other, however independent their work (measured here: a key-preparation stream that happened
to share the compute stream's queue made the step 0.35 ms SLOWER — all of the cross-stream waits, none of the overlap).
`pick()` therefore measures: it times pairs of spinning launches (usc_spin) on a candidate and on every stream the
candidate has to run beside, and returns the first candidate that overlaps with all of them.

NEVER a high-priority stream.  Their queues are a separate set that the null stream cannot share, which made them the
obvious first candidates — and with ONE high-priority stream in use (the decoder's key preparation, a few small launches
per pass) every kernel on the normal-priority queues ran 2-5x slower: 63.8 ms per step instead of 24.3, backbone forward
14.9 instead of 6.4 ms, decoder forward 11.7 instead of 2.3 (tools/ab.sh, round 5; `profiles/r05_stream_priority.txt`).  The
first version of this probe only escaped it because first-use noise made it reject the high-priority candidate.

(The same slow mode appears with GPU_MAX_HW_QUEUES=8: 55.8 ms per step.  The default four queues per process are what this
module works with.)

No reference counterpart (the reference is single-stream PyTorch); used by datasets/prefetch.py and models/mask3d.py.
"""
from __future__ import annotations

import os
import time

import torch

from ._lib import check, lib

SPIN_US = 60
SPIN_N = 8
_REPS = 4
_PICKED = {}          # (device index, role) -> stream
REPORT = []           # one dict per pick(): what was measured (tools/stream_queue_probe.py prints it)


def overlap_ratio(a: torch.cuda.Stream, b: torch.cuda.Stream, us: int = SPIN_US, n: int = SPIN_N, reps: int = _REPS,
                  during=None) -> float:
    """device time of n spins on stream a while n more are issued on b (alternately) / n spins on a alone: ~1 when the
    two streams run beside each other, ~2 when they share a hardware queue.  Timed with events on a (host jitter cannot
    shorten it) and the MINIMUM of `reps` repetitions of each form is used: a repetition can only be measured too long (a
    preempted host thread, a busy device), never too short.  Synchronises the device.  `during`: called right before the
    two-stream form's spins are issued (e.g. to put an asynchronous collective in flight); what it returns is `.wait()`ed
    for after the measurement."""
    def run(second):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pending = during() if (during is not None and second is not None) else None
        e0.record(a)
        for _ in range(n):
            check(lib.usc_spin(us, 1, a.cuda_stream), "usc_spin")
            if second is not None:
                check(lib.usc_spin(us, 1, second.cuda_stream), "usc_spin")
        e1.record(a)
        if pending is not None:
            pending.wait()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    run(b)                                   # first use of a stream binds its queue: not timed
    one = min(run(None) for _ in range(reps))
    two = min(run(b) for _ in range(reps))
    return two / one


def pick(device, role: str, beside=(), max_candidates: int = 8) -> torch.cuda.Stream:
    """A stream for `role` on `device` that overlaps with the device's default stream, with every stream in `beside` and
    with the streams picked for other roles (cached per (device, role)).  Falls back to the best candidate seen, noted in
    REPORT, when none overlaps with all.  Candidates are NORMAL-priority streams only (see the module docstring)."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, role)
    if key in _PICKED:
        return _PICKED[key]
    if os.environ.get("USC3D_STREAM_PROBE", "1") == "0" or torch.cuda.is_current_stream_capturing():
        st = torch.cuda.Stream(device=device)
        _PICKED[key] = st
        return st
    with torch.cuda.device(idx):
        others = [torch.cuda.default_stream(idx)] + [s for s in beside if s is not None] \
            + [s for (d, r), s in _PICKED.items() if d == idx and r != role]
        best, best_worst, tried = None, 1e9, []
        for _ in range(max_candidates):
            cand = torch.cuda.Stream(device=device)
            if any(cand.cuda_stream == o.cuda_stream for o in others):
                continue
            ratios = [overlap_ratio(o, cand) for o in others]
            worst = max(ratios)
            tried.append({"ratios": [round(r, 2) for r in ratios]})
            if worst < best_worst:
                best, best_worst = cand, worst
            if worst < 1.35:
                break
    REPORT.append({"role": role, "device": idx, "shared_queue": best_worst >= 1.35, "tried": tried})
    _PICKED[key] = best
    return best


def recheck_under_collective(device, nbytes: int = 24 << 20) -> list:
    """After the process group exists and every role has its stream: does the weight-gradient lane (then the decoder's
    key-preparation stream) still run beside the compute stream while a gradient bucket's all-reduce is in flight?  RCCL
    brings a stream of its own; HIP keeps a process at four hardware queues per priority, so that stream shares a queue
    with one of ours — if the shared one makes a second stream serialise with the compute stream (ratio >= 1.35), that
    second stream is switched off: the lane through units.set_lane_max_rows(0), the key stream through
    models.mask3d.set_kv_side_stream(False).  -> one report dict per role (also appended to REPORT; bench.py prints them
    in config.streams).  Reference counterpart: none (DDP's bucket all-reduce, main_instance_segmentation.py:86-92, runs
    on NCCL's stream next to a single compute stream)."""
    import torch.distributed as dist
    out = []
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if not (dist.is_available() and dist.is_initialized()):
        return out
    with torch.cuda.device(idx):
        buf = torch.zeros(max(1, nbytes // 4), dtype=torch.float32, device=device)
        main = torch.cuda.default_stream(idx)
        for role in ("wgrad-lane", "keys"):
            st = _PICKED.get((idx, role))
            if st is None:
                continue
            # (issued from a third stream's context: the collective must not be ordered behind the spins on `main`)
            helper = _PICKED.get((idx, "prefetch")) or torch.cuda.Stream(device=device)

            def collective():
                with torch.cuda.stream(helper):
                    return dist.all_reduce(buf, async_op=True)
            ratio = overlap_ratio(main, st, during=collective)
            rec = {"role": role + " under an all-reduce", "device": idx, "shared_queue": ratio >= 1.35,
                   "tried": [{"ratios": [round(ratio, 2)]}], "switched_off": False}
            if ratio >= 1.35:
                rec["switched_off"] = True
                if role == "wgrad-lane":
                    from . import units
                    units.set_lane_max_rows(0)
                else:
                    from .models import mask3d
                    mask3d.set_kv_side_stream(False)
            REPORT.append(rec)
            out.append(rec)
        torch.cuda.synchronize()
    return out
