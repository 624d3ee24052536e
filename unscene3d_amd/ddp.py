"""Data-parallel gradient exchange: one process per GPU, one scene per
rank, gradients all-reduced over RCCL/xGMI.  The reference gets this implicitly from
PyTorch-Lightning DDP (main_instance_segmentation.py:86-92); here every parameter's
`.grad` is a view into ONE flat fp32 buffer (`flatten_grads`), exchanged either as a single
large all-reduce after backward (`all_reduce_mean_`) or in a few large buckets that start
while the backbone's backward is still running (`BucketedGradReducer`).  xGMI rings are
per-link bound: few, large collectives."""
from __future__ import annotations

import os

import torch


def flatten_grads(params):
    """Allocate one flat buffer and make each p.grad a view into it. Returns the buffer."""
    params = list(params)
    total = sum(p.numel() for p in params)
    flat = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
    off = 0
    for p in params:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p)
        off += n
    return flat


def all_reduce_mean_(flat, world_size: int):
    import torch.distributed as dist

    if world_size > 1:
        dist.all_reduce(flat)
        flat.div_(world_size)
    return flat


# USC3D_LANE_ORDERED_COLLECTIVES=0: the round-5 behaviour (the compute stream waits for the lane before every bucket)
LANE_ORDERED_COLLECTIVES = os.environ.get("USC3D_LANE_ORDERED_COLLECTIVES", "1") == "1"
# USC3D_EARLY_BUCKETS=0 (diagnostic): the reducer counts the reports but starts every bucket in finish()
_EARLY_BUCKETS = os.environ.get("USC3D_EARLY_BUCKETS", "1") == "1"


class BucketedGradReducer:
    """Overlaps the gradient all-reduce with backward.

    The flat buffer is cut into contiguous buckets of ~`bucket_bytes` (24 MB: large enough for the xGMI rings, small
    enough that the bucket mixing backbone and decoder parameters — reduced after backward — stays small) in parameter
    order.  The backward kernels of this
    package write parameter gradients straight into `p.grad` (ops.GRAD_IN_PLACE) and report each write through
    `ops.GRAD_WRITTEN_HOOK`; parameters that still go through autograd report through a post-accumulate hook.  The
    first step only LEARNS how many reports each parameter produces per step (shared decoder weights are written by
    twelve passes — eager kernels and graph replays (graphs.py) both report); from the second step on a bucket whose parameters all
    reached their learned count is reduced asynchronously while backward continues.

    Every rank issues the collectives in the same order whatever the timing: eligible buckets (every parameter reports)
    strictly from the last one to the first — the order backward finishes them — and the rest, in index order, in
    `finish()`.  A bucket that is not complete when backward ends is simply reduced in `finish()`, in that same order.

    The report counts may still depend on data-dependent paths (a skipped level, a pass that falls back to eager
    kernels with a different number of launches per parameter), so the ranks AGREE on the learned state after
    the first step: `expected` is the maximum over ranks and a bucket is eligible only if every one of its parameters
    reports on every rank with the same count.  A gradient write that arrives after its bucket's all-reduce was
    started (more reports than learned) would race with the collective; it is detected, exchanged between the ranks
    with that step's collectives and raised on every rank two steps later (`RuntimeError`; the flag is read from pinned
    host memory without a stream synchronisation).
    """

    def __init__(self, params, flat, world_size: int, bucket_bytes: int = 24 << 20):
        self.params = list(params)
        self.flat, self.world = flat, world_size
        self.bounds, self.members = [], []          # bucket -> (start, end) in flat, list of param indices
        self.bucket_of = {}
        off, start, cur = 0, 0, []
        for i, p in enumerate(self.params):
            cur.append(i)
            self.bucket_of[id(p)] = len(self.bounds)
            off += p.numel()
            if (off - start) * flat.element_size() >= bucket_bytes:
                self.bounds.append((start, off))
                self.members.append(cur)
                start, cur = off, []
        if cur:
            self.bounds.append((start, off))
            self.members.append(cur)
        for i in self.members[-1]:
            self.bucket_of[id(self.params[i])] = len(self.bounds) - 1
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.expected = None                         # learned reports per parameter and step
        self.counts = [0] * len(self.params)
        self.order, self.cursor = [], 0              # eligible buckets, last first
        self.handles, self.launched = [], []
        self.started_during_backward = 0
        self._late, self._late_flag, self._late_pending = False, None, []
        self._hooks = [p.register_post_accumulate_grad_hook(self.on_grad) for p in self.params]

    def install(self):
        from . import ops
        ops.GRAD_WRITTEN_HOOK = self.on_grad
        return self

    def begin_step(self):
        self.counts = [0] * len(self.params)
        self.handles, self.launched, self.cursor = [], [False] * len(self.bounds), 0

    def on_grad(self, param):
        i = self.index.get(id(param))
        if i is None:
            return
        self.counts[i] += 1
        if self.expected is not None:
            if self.launched and self.launched[self.bucket_of[id(param)]]:
                self._late = True        # the kernels of this write race with the bucket's collective already in flight
            from . import ops
            if _EARLY_BUCKETS and not (self.flat.is_cuda and ops.on_side_stream()):
                # (a report from the decoder's key-preparation stream only counts: a collective started here would be
                # ordered behind THAT stream alone; the next report on the compute stream, or finish(), starts it)
                self._advance()

    def _complete(self, b):
        return all(self.counts[i] >= self.expected[i] for i in self.members[b])

    def _collective_stream(self):
        """The stream a bucket's all-reduce is issued on: the decoder's key-preparation stream when there is one — it has a
        hardware queue of its own (streams.pick) and is idle while the backbone's backward pass runs — else the lane."""
        from . import ops, units
        dev = self.flat.device
        for st in ops.SIDE_STREAMS:
            if st.device == dev:
                return st
        ent = units._LANE.get(dev.index if dev.index is not None else torch.cuda.current_device())
        return None if ent is None else ent[0]

    def _launch(self, b):
        import torch.distributed as dist
        s, e = self.bounds[b]
        if self.flat.is_cuda:
            # A bucket's collective has to start behind the compute stream's gradient writes, the lane's (units.py) and the
            # key-preparation stream's.  Round 5 made the compute stream wait for the lane here (every bucket then
            # serialised the lane's work into the input-gradient chain: 26.1-27.5 ms per step against 23.9 with a one-rank
            # group); the first half of round 6 issued `all_reduce(async_op=True)` in the lane's order.  Both used
            # ProcessGroupNCCL's ASYNC path — its own stream and a Work object per collective — and with eight of those per
            # step every third to sixth step took 30-36 ms instead of 23.5 (`profiles/r06_world1_rccl_steps.txt`: two async
            # dummy collectives per step are enough to produce it, twenty synchronous ones are not).  A SYNCHRONOUS
            # collective runs on the stream it is called on (torch >= 2.8) — no fifth stream next to the rank's four
            # (more than four active hardware queues are time-sliced), no Work to retire: it is issued on a
            # second stream that waits for one event of the compute stream and one of the lane, and nobody waits for IT
            # until finish().
            from . import ops, units
            ops.join_side_streams()          # (the decoder's key-preparation stream writes lin_squeeze / in_proj gradients)
            cs = self._collective_stream() if LANE_ORDERED_COLLECTIVES else None
            if cs is not None:
                cur = torch.cuda.current_stream()
                lane_ev = units.lane_event(self.flat.device)          # (releases weight gradients the lane still holds)
                ev = torch.cuda.Event()
                ev.record(cur)
                cs.wait_event(ev)
                if lane_ev is not None:
                    cs.wait_event(lane_ev)
                with torch.cuda.stream(cs):
                    dist.all_reduce(self.flat[s:e])
                self._on_stream = cs
                self.launched[b] = True
                return
            units.join_lane(self.flat.device)
        dist.all_reduce(self.flat[s:e])
        self.launched[b] = True

    def _advance(self):
        while self.cursor < len(self.order) and self._complete(self.order[self.cursor]):
            self._launch(self.order[self.cursor])
            self.cursor += 1

    def finish(self):
        """After backward: reduce what has not been started (same order on every rank), wait, average."""
        import torch.distributed as dist
        self.started_during_backward = sum(self.launched)
        # Late-write flags of EARLIER steps, each copied to pinned host memory behind its all-reduce.  A .item() on the
        # device flag here would make the host wait for the whole backward pass before it can queue the remaining
        # buckets; instead the value of the step before last is read (its copy finished long ago: waiting for it
        # costs nothing and bounds the host's run-ahead at two steps), at the same step on every rank.
        lag = 1 if self.flat.is_cuda else 0       # host tensors (tests): nothing runs ahead, look at the last step
        self._check_late(lag)
        self._step_no = getattr(self, "_step_no", 0) + 1
        if self.expected is None:
            # agree across ranks: max count, and eligibility only where min == max > 0 on every rank
            cnt = torch.tensor(self.counts, dtype=torch.int64, device=self.flat.device)
            lo, hi = cnt.clone(), cnt.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            lo, hi = lo.tolist(), hi.tolist()
            self.expected = hi
            eligible = [b for b in range(len(self.bounds))
                        if all(lo[i] > 0 and lo[i] == hi[i] for i in self.members[b])]
            self.order = sorted(eligible, reverse=True)
        for b in self.order[self.cursor:]:
            self._launch(b)
        self.cursor = len(self.order)
        for b in range(len(self.bounds)):
            if not self.launched[b]:
                self._launch(b)
        if self._late_flag is None:
            self._late_flag = torch.zeros(1, dtype=torch.float32, device=self.flat.device)
        self._late_flag.fill_(1.0 if self._late else 0.0)
        self._late = False
        cs = getattr(self, "_on_stream", None)
        if cs is not None:                       # the buckets were reduced on a second stream: the compute stream joins it here
            torch.cuda.current_stream().wait_stream(cs)
            self._on_stream = None
        dist.all_reduce(self._late_flag, op=dist.ReduceOp.MAX)      # stream-ordered for RCCL; a 4-byte host wait for gloo
        self.flat.div_(self.world)
        if self._late_flag.is_cuda:
            # ONE pinned allocation for the reducer's lifetime, used as a ring (a slot is read two steps after it was
            # written).  Round 6: a fresh `torch.zeros(1).pin_memory()` per step asked the driver for pinned memory
            # again whenever the caching host allocator's block was not yet released — every third or fourth step under a
            # one-rank RCCL group — and a pinned-memory (un)map stalls the kernels in flight: 30-36 ms steps among 23.5 ms
            # ones, +1.7 ms on the mean (`profiles/r06_world1_rccl_ab.txt`).
            if getattr(self, "_late_ring", None) is None:
                self._late_ring = torch.zeros(16, dtype=torch.float32).pin_memory()
            slot = self._step_no % self._late_ring.numel()
            host = self._late_ring[slot:slot + 1]
            host.copy_(self._late_flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._late_pending.append((ev, host, self._step_no))
        else:
            self._late_pending.append((None, self._late_flag.clone(), self._step_no))
        return self.flat

    def _check_late(self, keep):
        while len(self._late_pending) > keep:
            ev, host, step_no = self._late_pending.pop(0)
            if ev is not None:
                ev.synchronize()
            if float(host[0]) > 0:
                raise RuntimeError(f"BucketedGradReducer: in step {step_no} of this reducer (the current one is "
                                   f"{getattr(self, '_step_no', 0) + 1}) some rank wrote a gradient after its bucket's "
                                   "all-reduce had started (more writes per step than learned in the first step); "
                                   "the gradients of that step — and the optimizer steps since — are unreliable: "
                                   "restore the last checkpoint, rebuild the reducer or set USC3D_OVERLAP_ALLREDUCE=0")

    def flush(self):
        """Look at the late-write flags of ALL finished steps, the most recent included (waits for their copies).  Call
        before writing a checkpoint and at the end of training: `finish()` alone reads the flags with a lag of one
        step (two on the device), so the flags of the last steps of a run would otherwise never be inspected.
        Collective-free; every rank holds the same (MAX-reduced) flags, so every rank raises or none does."""
        self._check_late(0)
