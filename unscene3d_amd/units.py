"""Native issue path of the backbone: "conv -> batch norm (+ residual) (+ ReLU)" units and whole residual blocks as
ONE autograd node each, every unit one C call each way (csrc/units.hip: usc_conv_bn_act_forward / _backward).

Why: the per-operator path (ops.conv_same + ops.batch_norm_act, one autograd Function and 1-3 ctypes calls per
operator) issued ~1 000 launches per step for the backbone from the interpreter at 10-20 us each, which put host
issue time level with device time.  Here a BasicBlock (reference models/modules/resnet_block.py:48-64)
is one Function: 2-3 native calls forward, 2-3 backward, and the residual branch's gradient is accumulated by the
input-gradient kernel itself (`dx_accumulate`) instead of a separate element-wise add per block.

Same kernels, same kernel choice and the same arithmetic as the per-operator path; that path stays for odd cases
(bias, non-default USC3D_CONV, an open profiler capture, CPU-side hooks on the parameters) and as the A/B
reference of the tests (USC3D_NATIVE_UNITS=0).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch

from . import ops
from . import profiler as _prof
from ._lib import BNDesc, KMap, check, lib

ENABLED = os.environ.get("USC3D_NATIVE_UNITS", "1") == "1"
# Weight gradients of the small maps forked onto a side stream (usc_set_side_stream).  Measured on the bench scene:
# 37.3 ms per step without, 39.0 ms with (20 k voxels: 33.2 vs 35.8) — two event records and two stream waits per
# unit cost the host and the device more than the overlap of two ~30 us launches returns.  Off by default.
FORK_WGRAD = os.environ.get("USC3D_FORK_WGRAD", "0") == "1"
# The weight-gradient lane (usc_set_wgrad_lane): weight gradients of maps up to LANE_MAX_ROWS rows are queued on a second
# stream and joined ONCE, at the end of the backward pass (and before a gradient bucket goes to a collective) — the
# input-gradient chain is not held up by them.  On for every map since round 5 (USC3D_WGRAD_LANE_MAX_ROWS=0 switches it
# off).  History worth keeping: rounds 3-4 measured the lane 0.4-0.9 ms per step SLOWER at every row bound and concluded
# that a second queue only competes with the chain; the lane's stream was a plain torch.cuda.Stream() then, and HIP had
# put it on the compute stream's hardware queue (streams.py) — all of the event traffic, none of the overlap.  On a stream
# measured to run beside the compute stream: 25.2 ms per step without the lane, 25.35 / 24.97 / 24.40 / 24.2 ms with the
# bound at 3 000 / 10 000 / 50 000 / all rows (bench scene, same box, alternating runs; loss bits equal up to 50 000 rows,
# above that the finest level's three 96 -> 96 gradients are no longer one grouped grid and sum in another order).
LANE_MAX_ROWS = int(os.environ.get("USC3D_WGRAD_LANE_MAX_ROWS", str(1 << 40)))
LANE_WS_BYTES = 192 << 20
# The lane's schedule inside the step program's backward pass (usc_wgrad_lane_hold, round 6): weight gradients of maps
# with >= LANE_HOLD_MIN_ROWS rows (the levels whose input gradients run on the tile-compacted kernel) are noted until
# the walk reaches a map with <= LANE_RELEASE_ROWS rows and run beside the coarse levels' latency-bound chain instead
# of beside the fine levels' input gradients.  USC3D_LANE_RELEASE_ROWS=0 switches the schedule off.
LANE_HOLD_MIN_ROWS = int(os.environ.get("USC3D_LANE_HOLD_MIN_ROWS", "24576"))
LANE_RELEASE_ROWS = int(os.environ.get("USC3D_LANE_RELEASE_ROWS", "3000"))
SAME, DOWN, UP = 0, 1, 2
# Grouped weight gradients (usc_spconv_wgrad_group): the stride-1 convolutions of one level's residual blocks have the
# same shape on the same kernel map; their weight gradients are off the backward pass's critical chain (nothing reads
# dW before the optimizer / the gradient exchange), so a unit only QUEUES (x, dy, dW) and the queue is flushed as ONE
# grid when the shape changes, when it is full, or when the backward pass ends.  On the 507- and 2 222-row levels of a
# 150 k-voxel scene a single weight gradient fills a quarter of the chip and needs a pair split plus a reduction
# launch; eleven of them in one grid need neither (round-3 probe: 378 -> 197 us for the eleven 256 -> 256 problems of the
# stride-16 level).  USC3D_GROUP_WGRAD=0 switches it off (A/B, and the bit-equality tests against the per-operator path).
GROUP_WGRAD = os.environ.get("USC3D_GROUP_WGRAD", "1") == "1"
_SIDE = {}     # device index -> torch.cuda.Stream handed to usc_set_side_stream
_LANE = {}     # device index -> (stream, scratch tensor) handed to usc_set_wgrad_lane, or None
_LANE_JOIN_QUEUED = {}     # device index -> id of the graph task whose end-of-backward lane join has been queued


def _lane_stream(device):
    """The lane's stream: plain, with a priority (USC3D_LANE_PRIORITY: torch's convention, lower = more urgent), or
    restricted to a subset of the CUs (USC3D_LANE_CU_PATTERN=<hex word repeated over the 256-CU mask>, e.g. 55555555 =
    every second CU) so that the main stream's latency-bound launches always find free CUs — experiment knobs."""
    pat = os.environ.get("USC3D_LANE_CU_PATTERN")
    if pat:
        hip = C.CDLL("libamdhip64.so")
        words = (C.c_uint32 * 8)(*([int(pat, 16) & 0xffffffff] * 8))
        raw = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(raw), 8, words)
        if rc != 0 or not raw.value:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
        return torch.cuda.ExternalStream(raw.value, device=device)
    pr = os.environ.get("USC3D_LANE_PRIORITY")
    if pr == "low":
        # HIP's lowest priority (torch offers normal and high only): the chain's kernels win every race for a free CU
        hip = C.CDLL("libamdhip64.so")
        lo, hi = C.c_int(), C.c_int()
        hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
        raw = C.c_void_p()
        rc = hip.hipStreamCreateWithPriority(C.byref(raw), 1, lo.value)          # 1 = hipStreamNonBlocking
        if rc != 0 or not raw.value:
            raise RuntimeError(f"hipStreamCreateWithPriority failed ({rc})")
        return torch.cuda.ExternalStream(raw.value, device=device)
    if pr is not None:
        return torch.cuda.Stream(device=device, priority=int(pr))
    from . import streams           # a stream measured to run beside the compute stream (streams.py)
    return streams.pick(device, "wgrad-lane")


def _lane(device):
    """(stream, scratch) of the device's weight-gradient lane, or None when it is switched off."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _LANE:
        if LANE_MAX_ROWS > 0 and not FORK_WGRAD:
            st = _lane_stream(device)
            ws = torch.empty(LANE_WS_BYTES, dtype=torch.uint8, device=device)
            check(lib.usc_set_wgrad_lane(st.cuda_stream, ws.data_ptr(), ws.numel(), LANE_MAX_ROWS), "usc_set_wgrad_lane")
            _LANE[key] = (st, ws)
        else:
            _LANE[key] = None
    return _LANE[key]


def set_lane_max_rows(max_rows: int):
    """Switch the lane's row bound at run time (0 = off): tests and A/B runs.  Joins and drops the current lanes; the
    next backward builds them again."""
    global LANE_MAX_ROWS
    for key, ent in list(_LANE.items()):
        if ent is not None:
            with torch.cuda.device(key):
                join_lane()
                torch.cuda.synchronize()
                check(lib.usc_set_wgrad_lane(None, None, 0, 0), "usc_set_wgrad_lane")
    _LANE.clear()
    LANE_MAX_ROWS = int(max_rows)


def join_lane(device=None):
    """The current stream waits for every weight gradient queued on the lane so far (no-op without a lane)."""
    key = torch.cuda.current_device() if device is None or device.index is None else device.index
    if _LANE.get(key) is not None:
        check(lib.usc_wgrad_lane_join(ops._stream()), "usc_wgrad_lane_join")


def lane_stream_behind_current(device=None):
    """The lane's stream after it has been made to wait for everything queued on the current stream so far (one event), or
    None without a lane.  Work issued on it from here on — a gradient bucket's collective (ddp.BucketedGradReducer) — is
    ordered behind the lane's weight gradients AND the current stream's, and the current stream waits for neither.
    Weight gradients still held by the lane's schedule (usc_wgrad_lane_hold) are released first."""
    key = torch.cuda.current_device() if device is None or device.index is None else device.index
    ent = _LANE.get(key)
    if ent is None:
        return None
    if lib.usc_wgrad_lane_holding():
        check(lib.usc_wgrad_lane_hold(0, 0, 0, ops._stream()), "usc_wgrad_lane_hold")
    ev = torch.cuda.Event()
    ev.record()
    ent[0].wait_event(ev)
    return ent[0]


def lane_event(device=None, release=True):
    """An event behind every weight gradient queued on the lane so far (held ones are released first unless release=False),
    or None without a lane: what a gradient bucket's collective on another stream has to wait for
    (ddp.BucketedGradReducer._launch), or the early optimizer step of parameters whose gradients are final."""
    key = torch.cuda.current_device() if device is None or device.index is None else device.index
    ent = _LANE.get(key)
    if ent is None:
        return None
    if release and lib.usc_wgrad_lane_holding():
        check(lib.usc_wgrad_lane_hold(0, 0, 0, ops._stream()), "usc_wgrad_lane_hold")
    ev = torch.cuda.Event()
    ev.record(ent[0])
    return ev


def _join_lane_after_backward(key):
    _LANE_JOIN_QUEUED.pop(key, None)
    with torch.cuda.device(key):
        join_lane()


def queue_lane_join(key):
    """Queue ONE lane join at the end of the running backward pass.  Keyed on the graph task (ops._graph_task): after a
    backward pass that raised — the engine drops its callbacks then — the next pass queues its own join again."""
    task = ops._graph_task()
    if task < 0:                       # not inside a backward pass (a direct call of a backward function): join now
        with torch.cuda.device(key):
            join_lane()
        return
    if _LANE_JOIN_QUEUED.get(key) != task:
        _LANE_JOIN_QUEUED[key] = task
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _join_lane_after_backward(key))


def _lane_hold(device, n_in, n_out, in_place_dW, *tensors):
    """Called by a unit's backward before the native call: when that call will queue its weight gradient on the lane,
    the tensors it reads there are kept away from the allocator until the lane is past them, and the join at the end
    of this backward pass is queued (once).  -> True when the lane will be used."""
    lane = _lane(device)
    if lane is None or not in_place_dW or n_in <= 0 or max(n_in, n_out) > LANE_MAX_ROWS:
        return False
    for t in tensors:
        t.record_stream(lane[0])
    key = device.index if device.index is not None else torch.cuda.current_device()
    queue_lane_join(key)
    return True


def _ensure_side_stream(device):
    """One side stream per device for the weight gradients of the small maps (csrc/units.hip: forked and joined inside
    each backward call, so the caller's stream order — allocator, collectives, graph capture — is untouched)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _SIDE:
        if FORK_WGRAD:
            st = torch.cuda.Stream(device=device)
            check(lib.usc_set_side_stream(st.cuda_stream), "usc_set_side_stream")
            _SIDE[key] = st
        else:
            _SIDE[key] = None


class KMapRef:
    """A usc_kmap plus the tensors it points into (kept alive for the batch)."""

    __slots__ = ("struct", "ref", "keep", "n_in", "n_out", "K")

    def __init__(self, nbr, perm, tmask, rb, n_in, n_out, K):
        s = KMap()
        s.nbr = None if nbr is None else nbr.data_ptr()
        s.perm = None if perm is None else perm.data_ptr()
        s.tile_mask = None if tmask is None else tmask.data_ptr()
        if rb is not None:
            s.pair_in, s.pair_out, s.koff = rb.in_idx.data_ptr(), rb.out_idx.data_ptr(), rb.koff.data_ptr()
            s.pair_capacity = rb.capacity
        s.n_in, s.n_out, s.K = n_in, n_out, K
        self.struct, self.ref = s, C.byref(s)
        self.keep = (nbr, perm, tmask, None if rb is None else (rb.in_idx, rb.out_idx, rb.koff))
        self.n_in, self.n_out, self.K = n_in, n_out, K


def kmap_from_table(nbr, rb, n_in):
    """Kernel map of a neighbour table i32[K, n_out] with its row order and pair lists."""
    perm, tmask = ops.rowsort(nbr) if (1 < nbr.shape[0] <= 32 and nbr.shape[1] > 0) else (None, None)
    return KMapRef(nbr, perm, tmask, rb, n_in, nbr.shape[1], nbr.shape[0])


def kmap_identity(n):
    return KMapRef(None, None, None, None, n, n, 1)


class _WgradQueue:
    """Deferred same-shape weight gradients of the running backward pass (one queue per device)."""

    def __init__(self):
        self.key, self.items, self.task = None, [], None

    def push(self, key, kmap, x, dy, W_param, dW):
        task = ops._graph_task()
        if self.task is not None and task != self.task:
            # left behind by a backward pass that raised (the engine drops its callbacks then): those gradients belong
            # to a pass that never finished
            self.items, self.key, self.task = [], None, None
        if self.key is not None and key != self.key:
            self.flush()
        self.key = key
        self.items.append((kmap, x, dy, W_param, dW))
        if len(self.items) >= _group_max() or task < 0:
            self.flush()
        elif self.task is None:
            self.task = task
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)

    def _end_of_backward(self):
        self.task = None
        self.flush()

    def flush(self):
        items, key = self.items, self.key
        self.items, self.key = [], None
        if not items:
            return
        kmap = items[0][0]
        cin, cout, K = key[1], key[2], kmap.K
        s = kmap.struct
        R = len(items)
        with torch.cuda.device(items[0][1].device):
            st = ops._stream()
            if R >= 2 and lib.usc_spconv_wgrad_group_ok(R, cin, cout, K, s.pair_capacity):
                pa = (C.c_void_p * R)(*[it[1].data_ptr() for it in items])
                pb = (C.c_void_p * R)(*[it[2].data_ptr() for it in items])
                pw = (C.c_void_p * R)(*[it[4].data_ptr() for it in items])
                check(lib.usc_spconv_wgrad_group(R, pa, pb, pw, cin, cout, K, s.pair_in, s.pair_out, s.koff,
                                                 s.pair_capacity, 1, st), "usc_spconv_wgrad_group")
            else:                                   # a lone problem (or too few for the grouped grid): the usual launch
                for _, x, dy, _, dW in items:
                    wsb = lib.usc_spconv_wgrad_ws_bytes_rows(K, cin, cout, s.pair_capacity)
                    ws = workspace(wsb, x.device)
                    check(lib.usc_spconv_wgrad(x.data_ptr(), cin, dy.data_ptr(), cout, K, s.pair_in, s.pair_out, s.koff,
                                               s.pair_capacity, dW.data_ptr(), 1, ws.data_ptr(), ws.numel(), st),
                          "usc_spconv_wgrad")
        ops._grad_written(*[it[3] for it in items])


_WGQ = {}
_GROUP_MAX = []


def _group_max():
    if not _GROUP_MAX:
        _GROUP_MAX.append(int(lib.usc_spconv_wgrad_group_max()))
    return _GROUP_MAX[0]


def _wgrad_queue(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    q = _WGQ.get(key)
    if q is None:
        q = _WGQ[key] = _WgradQueue()
    return q


def flush_deferred_wgrads(device=None):
    """Issue every weight gradient still queued (called by the engine at the end of a backward pass; call it yourself
    before reading .grad from inside a backward hook)."""
    for q in (_WGQ.values() if device is None else [_wgrad_queue(device)]):
        q.flush()


def _defer_wgrad(kmap, kind, cin, cout, in_place):
    """May this unit's weight gradient go to the queue?  Stride-1 table convolutions on the coarse levels whose
    gradient is added into p.grad in place, outside graph capture, lane and fork modes off."""
    if not (GROUP_WGRAD and in_place and kind == SAME and kmap.K > 1 and kmap.struct.pair_in and not FORK_WGRAD
            and LANE_MAX_ROWS == 0):
        return False
    return bool(lib.usc_spconv_wgrad_group_ok(2, cin, cout, kmap.K, kmap.struct.pair_capacity) or
                lib.usc_spconv_wgrad_group_ok(_group_max(), cin, cout, kmap.K, kmap.struct.pair_capacity))


# one grow-only scratch buffer per device for the native calls (all on the compute stream: stream order makes the
# reuse safe; the caching allocator hands a replaced buffer only to later work of the same stream)
_WS = {}


def workspace(nbytes: int, device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + (1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


# usc_bn descriptors per module, OUTSIDE the module: a ctypes structure inside `bn.__dict__` made copy.deepcopy(model)
# (EMA copies), torch.save(model) and pickling for spawn fail ("ctypes objects containing pointers cannot be pickled")
_BN_DESC = weakref.WeakKeyDictionary()


def _bn_desc(bn: torch.nn.BatchNorm1d, training: bool):
    """usc_bn of an nn.BatchNorm1d, cached per module while its tensors stay where they are."""
    key = (bn.weight.data_ptr(), bn.bias.data_ptr(), 0 if bn.running_mean is None else bn.running_mean.data_ptr(),
           training, bn.momentum, bn.eps)
    cached = _BN_DESC.get(bn)
    if cached is not None and cached[0] == key:
        return cached[1]
    d = BNDesc()
    d.gamma, d.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
    track = bn.track_running_stats and bn.running_mean is not None
    d.running_mean = bn.running_mean.data_ptr() if track else None
    d.running_var = bn.running_var.data_ptr() if track else None
    d.num_batches_tracked = bn.num_batches_tracked.data_ptr() if (track and training and
                                                                   bn.num_batches_tracked is not None) else None
    if bn.momentum is None:
        raise NotImplementedError("cumulative-average batch norm (momentum=None) is not on the hot path")
    d.eps, d.momentum, d.c, d.training = float(bn.eps), float(bn.momentum), bn.num_features, int(training)
    ref = (d, C.byref(d))
    _BN_DESC[bn] = (key, ref)
    return ref


def _training(bn):
    return bn.training or not bn.track_running_stats


def usable(x, conv, *more_convs):
    """The native path covers f32 HIP features, bias-free convs, the default kernel dispatch, no open profiler capture."""
    if not (ENABLED and ops.CONV_PATH == "sorted" and _prof._active is None):
        return False
    if not (x.is_cuda and x.dtype == torch.float32):
        return False
    for c in (conv,) + more_convs:
        if c is not None and c.bias is not None:
            return False
    return True


# --------------------------------------------------------------------------------------------------------------
# one unit each way (plain functions: the autograd nodes below compose them)
def unit_forward(x, W3, bn, kmap: KMapRef, kind, residual, relu):
    """-> (y conv output, stats f32[4,c], out).  W3 f32[K,cin,cout]."""
    dev = x.device
    K, cin, cout = W3.shape
    n_out = kmap.n_in if kind == UP else kmap.n_out
    y = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
    out = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
    stats = torch.empty((4, cout), dtype=torch.float32, device=dev)
    _, bref = _bn_desc(bn, _training(bn))
    wsb = lib.usc_unit_ws_bytes(kmap.ref, kind, cin, cout)
    ws = workspace(wsb, dev)
    check(lib.usc_conv_bn_act_forward(kmap.ref, kind, x.data_ptr(), cin, W3.data_ptr(), cout, bref,
                                      None if residual is None else residual.data_ptr(), int(relu), y.data_ptr(),
                                      stats.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), ops._stream()),
          "usc_conv_bn_act_forward")
    return y, stats, out


def unit_backward(x, W3, bn, kmap, kind, y, stats, out_relu, dout, dy_buf, want_dres, dx, dx_accumulate, need_dx,
                  W_param, g_param, b_param, defer_ok=False):
    """Backward of one unit.  dy_buf: scratch [n_out, cout] (may be shared between the units of a block — unless
    defer_ok: the weight gradient may then be queued for a grouped launch that reads dy_buf later).
    -> (dx or None, dres or None, dW, dgamma, dbeta) — the last three None when written into .grad in place."""
    dev = x.device
    K, cin, cout = W3.shape
    n_out, n_in = (kmap.n_in, kmap.n_out) if kind == UP else (kmap.n_out, kmap.n_in)
    dres = torch.empty((n_out, cout), dtype=torch.float32, device=dev) if want_dres else None
    if need_dx and dx is None:
        dx = torch.empty((n_in, cin), dtype=torch.float32, device=dev)
        dx_accumulate = False
    _ensure_side_stream(dev)
    tW = ops._grad_target(W_param)
    _lane_hold(dev, n_in, n_out, tW is not None, x, dy_buf)
    tg, tb = ops._grad_target(g_param), ops._grad_target(b_param)
    bn_in_place = tg is not None and tb is not None
    dW = tW if tW is not None else torch.empty(W_param.shape, dtype=torch.float32, device=dev)
    dg = tg if bn_in_place else torch.empty(cout, dtype=torch.float32, device=dev)
    db = tb if bn_in_place else torch.empty(cout, dtype=torch.float32, device=dev)
    _, bref = _bn_desc(bn, _training(bn))
    wsb = lib.usc_unit_ws_bytes(kmap.ref, kind, cin, cout)
    ws = workspace(wsb, dev)
    defer = defer_ok and _defer_wgrad(kmap, kind, cin, cout, tW is not None)
    check(lib.usc_conv_bn_act_backward(kmap.ref, kind, x.data_ptr(), cin, W3.data_ptr(), cout, bref, y.data_ptr(),
                                       stats.data_ptr(), None if out_relu is None else out_relu.data_ptr(),
                                       dout.data_ptr(), dy_buf.data_ptr(), None if dres is None else dres.data_ptr(),
                                       dx.data_ptr() if need_dx else None, int(bool(dx_accumulate)),
                                       None if defer else dW.data_ptr(),
                                       int(tW is not None), dg.data_ptr(), db.data_ptr(), int(bn_in_place),
                                       ws.data_ptr(), ws.numel(), ops._stream()), "usc_conv_bn_act_backward")
    if defer:
        # queued: x and this unit's OWN dy stay referenced until the grouped launch; the write is reported then
        _wgrad_queue(dev).push((id(kmap), cin, cout), kmap, x, dy_buf, W_param, tW)
        dW = None
    elif tW is not None:
        ops._grad_written(W_param)
        dW = None
    if bn_in_place:
        ops._grad_written(g_param, b_param)
        dg = db = None
    return (dx if need_dx else None), dres, dW, dg, db


def _w3(W):
    return W if W.dim() == 3 else W[None]


class _Unit(torch.autograd.Function):
    """out = [relu](BN(conv(x)) [+ residual]) as one autograd node."""

    @staticmethod
    def forward(ctx, x, W, gamma, beta, residual, bn, kmap, kind, relu):
        x = x.contiguous()
        res = None if residual is None else residual.contiguous()
        W3 = _w3(W).contiguous()
        y, stats, out = unit_forward(x, W3, bn, kmap, kind, res, relu)
        ctx.save_for_backward(x, W3, y, stats, out if relu else None)
        ctx.bn, ctx.kmap, ctx.kind, ctx.has_res = bn, kmap, kind, residual is not None
        ctx.params = (W, gamma, beta)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, W3, y, stats, out_relu = ctx.saved_tensors
        dout = dout.contiguous()
        W, gamma, beta = ctx.params
        dy = torch.empty_like(y)
        dx, dres, dW, dg, db = unit_backward(x, W3, ctx.bn, ctx.kmap, ctx.kind, y, stats, out_relu, dout, dy,
                                             ctx.has_res, None, False, ctx.needs_input_grad[0], W, gamma, beta,
                                             defer_ok=True)
        if dW is not None and W.dim() == 2:
            dW = dW.view(W.shape)
        return dx, dW, dg, db, dres, None, None, None, None


def conv_bn_act(x, conv_weight, bn, kmap, kind, residual=None, relu=True):
    return _Unit.apply(x, conv_weight, bn.weight, bn.bias, residual, bn, kmap, kind, relu)


class _BasicBlock(torch.autograd.Function):
    """relu(norm2(conv2(relu(norm1(conv1(x))))) + residual), residual = x or norm_d(conv_d(x))
    (reference models/modules/resnet_block.py:48-64, models/resnet.py:124-146) as one autograd node."""

    @staticmethod
    def forward(ctx, x, W1, g1, b1, W2, g2, b2, Wd, gd, bd, bns, kmap, kmap_id):
        x = x.contiguous()
        bn1, bn2, bnd = bns
        W1c, W2c = W1.contiguous(), W2.contiguous()
        y1, st1, a1 = unit_forward(x, W1c, bn1, kmap, SAME, None, True)
        saved_d = (None, None, None)
        if Wd is not None:
            Wdc = _w3(Wd).contiguous()
            yd, std, r = unit_forward(x, Wdc, bnd, kmap_id, SAME, None, False)
            saved_d = (Wdc, yd, std)
        else:
            r = x
        y2, st2, out = unit_forward(a1, W2c, bn2, kmap, SAME, r, True)
        ctx.save_for_backward(x, W1c, W2c, y1, st1, a1, y2, st2, out, *saved_d)
        ctx.bns, ctx.kmap, ctx.kmap_id = bns, kmap, kmap_id
        ctx.params = (W1, g1, b1, W2, g2, b2, Wd, gd, bd)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, W1c, W2c, y1, st1, a1, y2, st2, out, Wdc, yd, std = ctx.saved_tensors
        W1, g1, b1, W2, g2, b2, Wd, gd, bd = ctx.params
        bn1, bn2, bnd = ctx.bns
        dout = dout.contiguous()
        need_dx = ctx.needs_input_grad[0]
        dy = torch.empty_like(y2)                 # d(conv output) scratch, shared by the units (same shape) ...
        # ... unless their weight gradients go to the lane: each then reads its own dy after this function has moved on
        lane_on = _lane(x.device) is not None and x.shape[0] <= LANE_MAX_ROWS
        # ... or to the grouped launch (queued weight gradients read their dy when the queue is flushed)
        own_dy = lane_on or (GROUP_WGRAD and _defer_wgrad(ctx.kmap, SAME, W2c.shape[1], W2c.shape[2],
                                                          ops._grad_target(W2) is not None))
        dy1 = torch.empty_like(y2) if own_dy else dy
        dyd = torch.empty_like(y2) if (own_dy and Wdc is not None) else dy     # (never a queued unit's buffer)
        # unit 2: dout -> (d a1, d residual)
        da1, dres, dW2, dg2, db2 = unit_backward(a1, W2c, bn2, ctx.kmap, SAME, y2, st2, out, dout, dy, True, None,
                                                 False, True, W2, g2, b2, defer_ok=own_dy)
        dWd = dgd = dbd = None
        if Wdc is None:
            # identity residual: conv1's input gradient is accumulated straight onto the residual gradient
            dx, _, dW1, dg1, db1 = unit_backward(x, W1c, bn1, ctx.kmap, SAME, y1, st1, a1, da1, dy1, False, dres, True,
                                                 need_dx, W1, g1, b1, defer_ok=own_dy)
        else:
            dx, _, dW1, dg1, db1 = unit_backward(x, W1c, bn1, ctx.kmap, SAME, y1, st1, a1, da1, dy1, False, None, False,
                                                 need_dx, W1, g1, b1, defer_ok=own_dy)
            dx, _, dWd, dgd, dbd = unit_backward(x, Wdc, bnd, ctx.kmap_id, SAME, yd, std, None, dres, dyd, False, dx,
                                                 True, need_dx, Wd, gd, bd)
            if dWd is not None and Wd.dim() == 2:
                dWd = dWd.view(Wd.shape)
        return dx, dW1, dg1, db1, dW2, dg2, db2, dWd, dgd, dbd, None, None, None


def basic_block(x, block, kmap, kmap_id):
    """`block`: a BasicBlock module (conv1/norm1/conv2/norm2/downsample)."""
    ds = block.downsample
    if ds is not None:
        Wd, gd, bd, bnd = ds[0].kernel, ds[1].bn.weight, ds[1].bn.bias, ds[1].bn
    else:
        Wd = gd = bd = bnd = None
    bn1, bn2 = block.norm1.bn, block.norm2.bn
    return _BasicBlock.apply(x, block.conv1.kernel, bn1.weight, bn1.bias, block.conv2.kernel, bn2.weight, bn2.bias,
                             Wd, gd, bd, (bn1, bn2, bnd), kmap, kmap_id)
