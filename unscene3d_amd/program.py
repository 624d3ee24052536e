"""The whole Res16UNet trunk as ONE step program each way (csrc/units.hip: usc_program_run).

The reference walks `Res16UNetBase.forward` (models/res16unet.py:224-297) module by module from the interpreter and
autograd walks it back node by node.  units.py already made a residual block one autograd node and a unit one C call;
what was left on the host per training step were ~190 Python-issued native calls, ~450 small tensor allocations and 33
autograd nodes for the trunk: 3 ms forward and 3.5 ms backward of the 13 ms the host needs to issue a 150 k-voxel step
(tools/host_profile.py), and most of the 15.5 ms of a 20 k-voxel step, which is host-bound.

Here the trunk is ONE autograd node:
* the walk is planned once per model (`_plan`: the units, their inputs / residuals / outputs as symbolic buffers, the
  skip concatenations) — the topology of `_trunk`, nothing else;
* per forward pass ONE arena holds every activation (conv outputs, normalised outputs, BN statistics, concatenations);
  the steps are filled in with raw pointers and `usc_program_run` launches the 62 units and 4 concatenations from C;
* the backward pass is the reverse walk over the same plan: ONE gradient arena, the units' backward steps (parameter
  gradients added into p.grad in place, weight gradients of a level's same-shape convolutions queued and issued as
  grouped launches inside the library), column splits for the concatenations and fan-in adds where a tensor has several
  consumers; it is cut into one C call per U-Net stage so that a gradient reducer hears about finished parameters
  while the earlier stages are still being issued (ops.GRAD_WRITTEN_HOOK).

Same kernels and the same arithmetic per unit as units.py (`usc_conv_bn_act_forward/backward`); features are bit-equal
to that path, gradients equal to rounding (different fan-in order).  Falls back to the per-block path whenever a piece
is not the plain form: a bias, a Bottleneck block, a parameter without an in-place gradient target, an open profiler
capture, `USC3D_BACKBONE_PROGRAM=0`.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import ops, units
from ._lib import STEP_ADD, STEP_CAT, STEP_SPLIT, STEP_UNIT_BWD, STEP_UNIT_FWD, Step, check, lib

ENABLED = os.environ.get("USC3D_BACKBONE_PROGRAM", "1") == "1"
SAME, DOWN, UP = units.SAME, units.DOWN, units.UP
_ALIGN = 64          # floats (256 bytes)


class _Plan:
    """Symbolic walk of the trunk: `ops` in forward order, `bufs` id -> (level, channels), `levels` = ids of the five
    block outputs [s16, s8, s4, s2, s1], `act_floats[l]` / `grad_floats[l]` = floats per row of level l the activation /
    gradient arenas need, `n_bufs` for the alignment slack."""

    def __init__(self):
        self.ops, self.bufs, self.levels = [], {}, []
        self.input = None
        self.params = []

    def new(self, level, c):
        i = len(self.bufs)
        self.bufs[i] = (level, c)
        return i


def _plan(model):
    """-> _Plan, or None when the trunk contains something the step program does not cover."""
    from .MinkowskiEngine import MinkowskiBatchNorm, MinkowskiConvolution, MinkowskiConvolutionTranspose
    from .models.modules.resnet_block import BasicBlockBase
    pl = _Plan()

    def unit(stage, conv, norm, kind, level, x, res, relu):
        if not (conv.bias is None and isinstance(norm, MinkowskiBatchNorm) and norm.bn.affine
                and norm.bn.momentum is not None):
            raise NotImplementedError
        if kind == SAME and not (isinstance(conv, MinkowskiConvolution) and conv.stride == 1):
            raise NotImplementedError
        if kind == DOWN and not (isinstance(conv, MinkowskiConvolution) and conv.stride == 2 and conv.ksize == 2):
            raise NotImplementedError
        if kind == UP and not (isinstance(conv, MinkowskiConvolutionTranspose) and conv.stride == 2 and conv.ksize == 2):
            raise NotImplementedError
        lout = level + 1 if kind == DOWN else (level - 1 if kind == UP else level)
        y, out = pl.new(lout, conv.out_channels), pl.new(lout, conv.out_channels)
        pl.ops.append(dict(t="unit", stage=stage, conv=conv, bn=norm.bn, kind=kind, lin=level, lout=lout, x=x, res=res,
                           relu=relu, y=y, out=out, cin=conv.in_channels, cout=conv.out_channels,
                           ksize=conv.ksize, kvol=conv.kernel_volume))
        return out

    def blocks(stage, seq, level, x):
        for blk in seq:
            if not isinstance(blk, BasicBlockBase) or blk.conv1.stride != 1 or blk.conv2.stride != 1 \
                    or blk.conv1.kernel_volume == 1 or blk.conv1.ksize != blk.conv2.ksize:
                raise NotImplementedError
            a1 = unit(stage, blk.conv1, blk.norm1, SAME, level, x, None, True)
            r = x
            if blk.downsample is not None:
                ds = blk.downsample
                if len(ds) != 2 or ds[0].kernel_volume != 1:
                    raise NotImplementedError
                r = unit(stage, ds[0], ds[1], SAME, level, x, None, False)
            x = unit(stage, blk.conv2, blk.norm2, SAME, level, a1, r, True)
        return x

    try:
        pl.input = pl.new(0, model.conv0p1s1.in_channels)
        x = unit("conv0p1s1", model.conv0p1s1, model.bn0, SAME, 0, pl.input, None, True)
        skip = [x]
        for i, (cname, nname, bname) in enumerate(model._DOWN):
            x = unit(cname, getattr(model, cname), getattr(model, nname), DOWN, i, x, None, True)
            x = blocks(bname, getattr(model, bname), i + 1, x)
            skip.append(x)
        pl.levels = [x]
        top = len(model._DOWN)
        for j, (cname, nname, bname) in enumerate(model._UP):
            u = unit(cname, getattr(model, cname), getattr(model, nname), UP, top - j, x, None, True)
            s = skip[top - 1 - j]
            ca, cb = pl.bufs[u][1], pl.bufs[s][1]
            if ca % 4 or cb % 4:
                raise NotImplementedError
            c = pl.new(top - 1 - j, ca + cb)
            pl.ops.append(dict(t="cat", stage=cname, a=u, b=s, out=c, level=top - 1 - j, ca=ca, cb=cb))
            x = blocks(bname, getattr(model, bname), top - 1 - j, c)
            pl.levels.append(x)
    except (NotImplementedError, AttributeError):
        return None
    n_levels = top + 1
    pl.n_levels = n_levels
    act, grad = [0] * n_levels, [0] * n_levels
    n_act = n_grad = 0
    for op in pl.ops:
        if op["t"] == "unit":
            act[op["lout"]] += 2 * op["cout"]
            n_act += 3
            grad[op["lout"]] += 2 * op["cout"]          # dy, dres
            grad[op["lin"]] += 2 * op["cin"]            # dx (+ a temporary when the target cannot be accumulated into)
            n_grad += 4
        else:
            act[op["level"]] += op["ca"] + op["cb"]
            n_act += 1
            grad[op["level"]] += op["ca"] + op["cb"]
            n_grad += 2
    pl.act_floats, pl.grad_floats, pl.n_act, pl.n_grad = act, grad, n_act, n_grad
    pl.stat_floats = sum(4 * op["cout"] for op in pl.ops if op["t"] == "unit")
    seen, params = set(), []
    for op in pl.ops:
        if op["t"] == "unit":
            for p in (op["conv"].kernel, op["bn"].weight, op["bn"].bias):
                if id(p) not in seen:
                    seen.add(id(p))
                    params.append(p)
    pl.params = params
    return pl


def _module_key(model):
    """Identity of every module the plan references: a conversion that swaps modules after the first forward pass (a
    batch-norm conversion, a re-built block) must not be served the old plan, which holds the old modules' parameters.
    (The cached plan keeps references to the modules it uses, so none of THEIR ids can be handed to a new object while
    it is cached; ~30 us per call for the few hundred modules of the trunk — round-5 advice, left as it is.)"""
    return tuple(id(m) for m in model.modules())


def plan_of(model):
    cached = model.__dict__.get("_usc_program_plan")
    key = _module_key(model)
    if cached is None or cached[1] != key:
        cached = model.__dict__["_usc_program_plan"] = (_plan(model) or False, key)
    return cached[0] or None


class _Carver:
    """Hands out 256-byte aligned float ranges of one arena as raw device pointers."""

    def __init__(self, floats, device):
        self.arena = torch.empty(int(floats), dtype=torch.float32, device=device)
        self.base, self.off, self.cap = self.arena.data_ptr(), 0, int(floats)

    def take(self, n):
        off = self.off
        self.off = (off + int(n) + _ALIGN - 1) // _ALIGN * _ALIGN
        if self.off > self.cap:
            raise RuntimeError("step program: arena too small (planning bug)")
        return self.base + 4 * off, off

    def view(self, off, n, c):
        return self.arena[off:off + n * c].view(n, c)


class _Pool:
    """Gradient buffers of one backward pass: chunks allocated on demand, exact-size reuse of released ranges.  The
    steps run in order on ONE stream, so a range released after the step that last reads it may be handed to any later
    step (the weight gradients queued inside a usc_program_run call read their dy until that call returns: those are
    released when the stage's call is closed).  Peak ~1.5 GB for a 150 k-voxel scene instead of the 8 GB the sum of
    all gradient buffers would take."""

    def __init__(self, device, chunk_floats):
        self.device, self.chunk_floats = device, int(chunk_floats)
        self.chunks, self.base, self.off, self.cap = [], 0, 0, 0
        self.free = {}

    def take(self, n):
        n = (int(n) + _ALIGN - 1) // _ALIGN * _ALIGN
        lst = self.free.get(n)
        if lst:
            return lst.pop()
        if self.off + n > self.cap:
            self.cap = max(self.chunk_floats, n)
            t = torch.empty(self.cap, dtype=torch.float32, device=self.device)
            self.chunks.append(t)
            self.base, self.off = t.data_ptr(), 0
        p = self.base + 4 * self.off
        self.off += n
        return p

    def release(self, p, n):
        n = (int(n) + _ALIGN - 1) // _ALIGN * _ALIGN
        self.free.setdefault(n, []).append(p)


def usable(model, x):
    """May this forward pass run as a step program?"""
    if not (ENABLED and units.usable(x.F, None) and x.F.shape[0] > 0):
        return None
    pl = plan_of(model)
    if pl is None:
        return None
    if torch.is_grad_enabled():
        need = [p.requires_grad for p in pl.params]
        if any(need):
            # the backward walk writes EVERY unit's parameter gradients in place and returns no input gradient: a
            # partly frozen trunk, a parameter without a gradient buffer or an input that wants its own gradient go
            # through the per-block path, which hands gradients back through autograd
            if not all(need) or x.F.requires_grad:
                return None
            for p in pl.params:
                if ops._grad_target(p) is None:
                    return None
    return pl


class _Trunk(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, pl, cm, ts, feats, *params):
        dev = feats.device
        rows = [cm.coord_map(ts << l).n for l in range(pl.n_levels)]
        floats = sum(r * f for r, f in zip(rows, pl.act_floats)) + pl.stat_floats + (pl.n_act + 8) * _ALIGN
        car = _Carver(floats, dev)
        ptr, offs = {pl.input: feats.data_ptr()}, {}
        steps = (Step * len(pl.ops))()
        kcache = {}

        def kmap(op):
            key = (op["kind"], op["lin"], op["kvol"])
            k = kcache.get(key)
            if k is None:
                t = ts << op["lin"]
                if op["kind"] == SAME:
                    k = cm.kmap_identity(t) if op["kvol"] == 1 else cm.kmap_cube(t, op["ksize"])
                elif op["kind"] == DOWN:
                    k = cm.kmap_down(t)
                else:
                    k = cm.kmap_down(t >> 1)
                kcache[key] = k
            return k

        stats_ptr, kmaps, bnrefs = [], [], []
        for i, op in enumerate(pl.ops):
            st = steps[i]
            if op["t"] == "unit":
                n_out, cout = rows[op["lout"]], op["cout"]
                py, offs[op["y"]] = car.take(n_out * cout)
                po, offs[op["out"]] = car.take(n_out * cout)
                ps, _ = car.take(4 * cout)
                ptr[op["y"]], ptr[op["out"]] = py, po
                km = kmap(op)
                bdesc, _ = units._bn_desc(op["bn"], units._training(op["bn"]))
                bref = C.addressof(bdesc)                  # (the descriptor is kept alive by units._BN_DESC and `bnrefs`)
                st.op, st.kind, st.cin, st.cout, st.relu = STEP_UNIT_FWD, op["kind"], op["cin"], cout, int(op["relu"])
                st.map, st.bn = C.addressof(km.struct), bref
                st.x, st.W = ptr[op["x"]], op["conv"].kernel.data_ptr()
                st.residual = ptr[op["res"]] if op["res"] is not None else None
                st.y, st.stats, st.out = py, ps, po
                stats_ptr.append(ps)
                kmaps.append(km)
                bnrefs.append((bref, bdesc))
            else:
                n = rows[op["level"]]
                pc, offs[op["out"]] = car.take(n * (op["ca"] + op["cb"]))
                ptr[op["out"]] = pc
                st.op, st.a, st.b, st.dst, st.n, st.ca, st.cb = STEP_CAT, ptr[op["a"]], ptr[op["b"]], pc, n, op["ca"], op["cb"]
                stats_ptr.append(None)
                kmaps.append(None)
                bnrefs.append(None)
        wsb = lib.usc_program_ws_bytes(steps, len(pl.ops))
        ws = units.workspace(wsb, dev)
        check(lib.usc_program_run(steps, 0, len(pl.ops), ws.data_ptr(), ws.numel(), ops._stream()), "usc_program_run")
        outs = tuple(car.view(offs[b], rows[pl.bufs[b][0]], pl.bufs[b][1]) for b in pl.levels)
        ctx.state = (model, pl, cm, ts, feats, car, ptr, stats_ptr, kmaps, bnrefs, rows)
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        model, pl, cm, ts, feats, car, ptr, stats_ptr, kmaps, bnrefs, rows = ctx.state
        dev = feats.device
        gar = _Pool(dev, max(rows[0] * 256, 1 << 20))
        lane = units._lane(dev)                  # weight-gradient lane (on by default): its launches read x and dy after
        if lane is not None:                     # this function has moved on -> nothing they read is re-used
            gar.release = lambda p_, n_: None
            car.arena.record_stream(lane[0])
        keep = [g.contiguous() if g is not None else None for g in gouts]
        G = {}                                   # buffer id -> [pointer, read-only (an incoming gradient tensor), floats]
        held_dy = []                             # dy buffers the current stage's queued weight gradients still read
        for b, g in zip(pl.levels, keep):
            if g is not None:
                G[b] = [g.data_ptr(), True, g.numel()]
        steps = (Step * (3 * len(pl.ops) + 8))()
        ns = 0
        segments, cur_params, stage = [], [], None       # one C call per U-Net stage: (end step, parameters it finished)

        def add_step(dst, src, n):
            nonlocal ns
            st = steps[ns]
            ns += 1
            st.op, st.dst, st.a, st.n = STEP_ADD, dst, src, n

        def contribute(bid, numel, can_accumulate):
            """Where the next contribution to the gradient of buffer `bid` goes -> (pointer, accumulate flag, finish):
            finish() is called after the producing step has been emitted (fan-in add when the target could not be
            accumulated into: an incoming gradient tensor is never written, and the pair-list forms only write)."""
            cur = G.get(bid)
            if cur is None:
                p = gar.take(numel)
                G[bid] = [p, False, numel]
                return p, 0, None
            if can_accumulate and not cur[1]:
                return cur[0], 1, None
            p = gar.take(numel)
            if cur[1]:                           # new buffer = this contribution + the incoming gradient; ours from now on
                def fin(p=p, old=cur[0]):
                    add_step(p, old, numel)
                    G[bid] = [p, False, numel]
            else:
                def fin(p=p, old=cur[0]):
                    add_step(old, p, numel)
                    gar.release(p, numel)
            return p, 0, fin

        def consumed(bid):
            """The gradient of buffer `bid` has been read by its producer's backward step: give the range back."""
            cur = G.pop(bid, None)
            if cur is not None and not cur[1]:
                gar.release(cur[0], cur[2])

        for i in range(len(pl.ops) - 1, -1, -1):
            op = pl.ops[i]
            if op["stage"] != stage:
                if stage is not None:
                    segments.append((ns, cur_params))
                    cur_params = []
                    for p_, n_ in held_dy:       # the stage's C call flushes its queued weight gradients before it returns
                        gar.release(p_, n_)
                    held_dy = []
                stage = op["stage"]
            if op["t"] == "cat":
                g = G.get(op["out"])
                if g is None:
                    continue
                n = rows[op["level"]]
                pa, _, fa = contribute(op["a"], n * op["ca"], False)
                cur_b = G.get(op["b"])
                if cur_b is not None and not cur_b[1]:
                    pb, acc_b, fb = cur_b[0], 1, None
                else:
                    pb, acc_b, fb = contribute(op["b"], n * op["cb"], False)
                st = steps[ns]
                ns += 1
                st.op, st.a, st.dst, st.dst2, st.n, st.ca, st.cb, st.accumulate = STEP_SPLIT, g[0], pa, pb, n, op["ca"], \
                    op["cb"], acc_b
                for f in (fa, fb):
                    if f is not None:
                        f()
                consumed(op["out"])
                continue
            g = G.get(op["out"])
            if g is None:
                continue                          # nothing downstream asked for this unit's gradient
            n_out, n_in, cin, cout = rows[op["lout"]], rows[op["lin"]], op["cin"], op["cout"]
            pdy = gar.take(n_out * cout)
            pres, fres = None, None
            if op["res"] is not None:
                pres, _, fres = contribute(op["res"], n_out * cout, False)
            need_dx = op["x"] != pl.input
            pdx, acc, fdx = (None, 0, None)
            if need_dx:
                pdx, acc, fdx = contribute(op["x"], n_in * cin, op["kind"] != DOWN)
            conv, bn = op["conv"], op["bn"]
            tW, tg, tb = ops._grad_target(conv.kernel), ops._grad_target(bn.weight), ops._grad_target(bn.bias)
            if tW is None or tg is None or tb is None:
                raise RuntimeError("step program: a parameter lost its in-place gradient target between forward and "
                                   "backward (keep p.grad allocated: optimizer.zero_grad(set_to_none=False))")
            st = steps[ns]
            ns += 1
            st.op, st.kind, st.cin, st.cout = STEP_UNIT_BWD, op["kind"], cin, cout
            st.map, st.bn = C.addressof(kmaps[i].struct), bnrefs[i][0]
            st.x, st.W = ptr[op["x"]], conv.kernel.data_ptr()
            st.y, st.stats = ptr[op["y"]], stats_ptr[i]
            st.out = ptr[op["out"]] if op["relu"] else None
            st.dout, st.dy, st.dres, st.dx, st.dx_accumulate = g[0], pdy, pres, pdx, acc
            st.dW, st.dW_accumulate, st.dgamma, st.dbeta, st.dbn_accumulate = tW.data_ptr(), 1, tg.data_ptr(), \
                tb.data_ptr(), 1
            st.defer_wgrad = int(units.GROUP_WGRAD)
            cur_params.extend((conv.kernel, bn.weight, bn.bias))
            for f in (fres, fdx):
                if f is not None:
                    f()
            consumed(op["out"])
            if st.defer_wgrad and op["kind"] == SAME and op["kvol"] > 1:
                held_dy.append((pdy, n_out * cout))
            else:
                gar.release(pdy, n_out * cout)
        segments.append((ns, cur_params))
        wsb = lib.usc_program_ws_bytes(steps, ns)
        ws = units.workspace(wsb, dev)
        stream = ops._stream()
        begin = 0
        # the lane's schedule (usc_wgrad_lane_hold): the fine decoder stages' weight gradients are noted, not launched,
        # until the walk reaches a stage whose maps cannot fill the chip; their parameters are reported then
        holding = lane is not None and units.LANE_HOLD_MIN_ROWS > units.LANE_RELEASE_ROWS > 0 and \
            any(min(rows[o["lin"]], rows[o["lout"]]) <= units.LANE_RELEASE_ROWS for o in pl.ops if o["t"] == "unit")
        unreported = []
        if holding:
            check(lib.usc_wgrad_lane_hold(1, units.LANE_HOLD_MIN_ROWS, units.LANE_RELEASE_ROWS, stream), "usc_wgrad_lane_hold")
        try:
            for end, plist in segments:
                if end > begin:
                    check(lib.usc_program_run(steps, begin, end, ws.data_ptr(), ws.numel(), stream), "usc_program_run")
                if holding and lib.usc_wgrad_lane_holding():
                    unreported.extend(plist)
                elif plist or unreported:
                    ops._grad_written(*unreported, *plist)   # finished stages are reported while the earlier ones are still issued
                    ops._params_final(unreported + plist)    # ... and are final: each unit's parameters are written once
                    unreported = []
                begin = end
            if holding:
                check(lib.usc_wgrad_lane_hold(0, 0, 0, stream), "usc_wgrad_lane_hold")
                if unreported:
                    ops._grad_written(*unreported)
                    ops._params_final(unreported)
        except BaseException:
            if holding:
                lib.usc_wgrad_lane_hold(-1, 0, 0, None)
            raise
        if lane is not None:
            for t in gar.chunks:
                t.record_stream(lane[0])
            key = dev.index if dev.index is not None else torch.cuda.current_device()
            units.queue_lane_join(key)
        ctx.state = None
        return (None, None, None, None, None) + (None,) * len(pl.params)


def trunk(model, x):
    """-> (list of the five block outputs as feature matrices [s16, s8, s4, s2, s1]) or None when the step program
    does not apply to this model / tensor / gradient configuration."""
    pl = usable(model, x)
    if pl is None:
        return None
    cm, ts = x.coordinate_manager, x._ts()
    feats = x.F.contiguous()
    return list(_Trunk.apply(model, pl, cm, ts, feats, *pl.params))
