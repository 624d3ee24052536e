"""HIP-graph capture of fixed-shape sub-networks that are replayed many times per step (the 12 decoder passes).

Differs from `torch.cuda.make_graphed_callables` in what matters for a module whose WEIGHTS ARE SHARED by all
captured callables (Mask3D's shared decoder, reference models/mask3d.py:100-156):

* parameter gradients are accumulated INSIDE the captured backward graph, straight into `p.grad`, by one
  multi-tensor add per pass.  Routed through autograd instead, every pass hands ~40 gradient tensors to
  AccumulateGrad, which issues one small add kernel each: 12 x 40 = 480 launches (~2 ms) per training step;
* an input that already lives in its static buffer is not copied again.

Requirements (checked at replay, RuntimeError otherwise): every parameter has a `.grad` tensor whose storage
does not change after capture — use `optimizer.zero_grad(set_to_none=False)` (or `ddp.flatten_grads`).
"""
from __future__ import annotations

import os

import torch

# "thread_local": other threads of the process (the RCCL watchdog polls events while a multi-GPU job captures) may
# keep issuing HIP calls during capture without invalidating it
_CAPTURE_MODE = "thread_local"


class _Runner:
    def __init__(self, module, sample_inputs, params):
        self.module = module
        self.params = params
        self.static_in = [torch.zeros_like(a).requires_grad_(a.requires_grad) for a in sample_inputs]
        self.grad_in_idx = [j for j, a in enumerate(self.static_in) if a.requires_grad]
        self.fwd_graph = torch.cuda.CUDAGraph()
        self.bwd_graph = torch.cuda.CUDAGraph()
        self.static_out = None
        self.static_grad_out = None
        self.static_grad_in = None
        self.grad_ptrs = None
        self.data_ptrs = None
        # chained input (capture_passes(chain_input=...)): position, the runner whose output buffer it aliases, and
        # whether THIS runner's output of the current step may still be read by autograd consumers (replayed with
        # gradients on, backward not yet run)
        self.chain_pos = None
        self.chain_prev = None
        self.live = False
        self.shared_pos = ()

    def check_param_data(self):
        """The captured kernels hold each parameter's device address: a parameter whose storage moved after capture
        (module.to()/.float(), an optimizer that re-points p.data into a flat buffer, load_state_dict(assign=True))
        would be read from freed memory while the optimizer updates storage the graphs never see."""
        for p, ptr in zip(self.params, self.data_ptrs):
            if p.data_ptr() != ptr:
                raise RuntimeError(
                    "graphed decoder pass: a parameter's storage moved after capture; build the optimizer (FlatAdamW "
                    "re-points p.data) and finish .to()/.float() BEFORE enable_decoder_graphs(), or call "
                    "disable_decoder_graphs() and capture again")

    def check_param_grads(self):
        self.check_param_data()
        for p, ptr in zip(self.params, self.grad_ptrs):
            if p.grad is None or p.grad.data_ptr() != ptr:
                raise RuntimeError(
                    "graphed decoder pass: a parameter's .grad was freed or reallocated after capture; keep gradient "
                    "buffers alive (optimizer.zero_grad(set_to_none=False) or ddp.flatten_grads) or disable the graphs")


# data pointer of a captured pass's OUTPUT buffer -> the static buffer its backward graph reads the output gradient from.
# The operator that produces that gradient (the decoder's LayerNorm over the pass output, ops._LayerNorm.backward) asks
# here and writes straight into the buffer: no copy in front of the backward replay (12 per training step).
_GRAD_OUT_BUFFERS = {}
STATS = {"grad_buffer_hits": 0, "grad_out_copies": 0}      # counted per backward replay (tests; tools)
_GRAD_BUFFER_PASSTHROUGH = os.environ.get("USC3D_GRAD_BUFFER_PASSTHROUGH", "1") == "1"


def grad_buffer_for(t):
    """-> the static output-gradient buffer of the captured pass whose output buffer `t` is (as stored, same element
    order as the pass output), or None.  `t` may be any contiguous VIEW of the output (the LayerNorm works on
    [rows, d]): what has to agree is the element count, the dtype and that both are dense in the same order — a
    buffer with permuted strides (several scenes per batch) is not handed out."""
    if not _GRAD_BUFFER_PASSTHROUGH:
        return None
    hit = _GRAD_OUT_BUFFERS.get((t.device.index, t.data_ptr()))
    if (hit is None or hit.numel() != t.numel() or hit.dtype != t.dtype or not hit.is_contiguous()
            or not t.is_contiguous()):
        return None
    STATS["grad_buffer_hits"] += 1
    return hit


def forget_grad_buffers():
    _GRAD_OUT_BUFFERS.clear()


class _GraphedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, *inputs):
        runner.check_param_data()
        if runner.chain_prev is None:          # the first pass of a step: nothing of the previous step is in flight any more
            for r in getattr(runner, "group", ()):
                r.live = False
        for j, (s, a) in enumerate(zip(runner.static_in, inputs)):
            if s.data_ptr() == a.data_ptr():
                continue                       # the producer wrote straight into the buffer (or it is the previous pass's output)
            if j == runner.chain_pos and runner.chain_prev is not None and runner.chain_prev.live:
                # this input ALIASES the previous pass's output buffer, and that output was produced with gradients on
                # in this step: its autograd consumers (the decoder's LayerNorm saves it) still read the buffer in
                # backward — copying another tensor over it would silently corrupt their gradients
                raise RuntimeError("graphed decoder pass: the chained input is not the previous pass's output, whose "
                                   "buffer is still needed by this step's backward; run this pass eagerly "
                                   "(Mask3D._eager_pass) or capture the passes without chain_input")
            hold = getattr(s, "_usc_holds", None)
            if hold is not None and hold[0] is a and hold[1] == a._version:
                continue                       # a buffer SHARED by several passes already holds this very tensor
            s.copy_(a)
            if hold is not None:
                s._usc_holds = (a, a._version)  # (keeps `a` alive: its storage cannot be handed to another tensor)
        runner.fwd_graph.replay()
        ctx.runner = runner
        # (grad mode is OFF inside Function.forward, so torch.is_grad_enabled() says nothing here — up to round 5 this
        #  flag was therefore never set and the chained-input guard above could not fire; whether a backward through this
        #  replay can follow is decided where the pass is CALLED (grad mode on and an input that requires a gradient;
        #  ctx.needs_input_grad ignores no_grad()).  Round 6 also measured the passes summing query_pos's
        #  gradient inside their backward graphs instead of autograd's 11 eager adds per step: same bits, 23.33 / 23.38 /
        #  23.35 vs 23.30 / 23.32 / 23.23 ms per step — an add is a kernel either way; not kept.)
        runner.live = bool(getattr(runner, "next_live", False))   # set by GraphedPass.__call__, OUTSIDE the Function
        return runner.static_out.detach()

    @staticmethod
    def backward(ctx, g):
        r = ctx.runner
        r.check_param_grads()
        if r.static_grad_out.data_ptr() != g.data_ptr():
            STATS["grad_out_copies"] += 1
            r.static_grad_out.copy_(g)
        r.bwd_graph.replay()
        r.live = False
        if r.chain_prev is None:               # the step's last backward replay: let go of the tensors the shared buffers held
            for j in r.shared_pos:
                if getattr(r.static_in[j], "_usc_holds", None) is not None:
                    r.static_in[j]._usc_holds = (None, -1)
        # the replayed graph added into every parameter's .grad: report it like the eager kernels do, so that a
        # gradient reducer never starts a bucket while replays that write into it are still to come (and so that
        # ranks running the same pass eagerly / graphed count the same number of writes)
        from . import ops
        ops._grad_written(*r.params)
        out = [None] * (1 + len(r.static_in))
        for j, gi in zip(r.grad_in_idx, r.static_grad_in):
            out[1 + j] = None if gi is None else gi.detach()
        return tuple(out)


class GraphedPass:
    def __init__(self, runner):
        self._runner = runner
        # the static input buffers as plain (detached) tensors: a producer may write its result straight into them —
        # an input whose data pointer equals its buffer's is not copied at replay
        self.input_buffers = [s.detach() for s in runner.static_in]

    def __call__(self, *inputs):
        self._runner.next_live = torch.is_grad_enabled() and any(isinstance(a, torch.Tensor) and a.requires_grad
                                                                 for a in inputs)
        return _GraphedFn.apply(self._runner, *inputs)


def capture_passes(modules, sample_inputs, warmup_iters: int = 3, shared_inputs=(), chain_input=None):
    """modules: callables (nn.Modules sharing parameters or not); sample_inputs: one tuple of tensors per module
    (requires_grad marks the inputs whose gradient is needed).  -> list of GraphedPass, replayed in the order
    given for the forward direction and in reverse for the backward one (like the captured order).
    shared_inputs: input positions that receive the SAME tensor in every pass of a step (the decoder's `query_pos`):
    the passes share one static buffer there and only the first replay of a step copies into it.
    chain_input: input position that receives the PREVIOUS pass's output (the decoder's `queries`): pass k+1 reads
    pass k's output buffer in place — no copy between consecutive replays."""
    assert len(modules) == len(sample_inputs)
    runners = []
    for m, smp in zip(modules, sample_inputs):
        params = [p for p in m.parameters() if p.requires_grad]
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        runners.append(_Runner(m, smp, params))
    for j in shared_inputs:
        first = runners[0].static_in[j]
        first._usc_holds = (None, -1)
        for r in runners[1:]:
            if r.static_in[j].shape != first.shape or r.static_in[j].dtype != first.dtype:
                raise RuntimeError("capture_passes: a shared input must have one shape in every pass")
            r.static_in[j] = first

    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for r in runners:   # eager warm-up (library handles, autotuning) — gradients discarded
            for _ in range(warmup_iters):
                out = r.module(*r.static_in)
                torch.autograd.grad((out,), [r.static_in[j] for j in r.grad_in_idx] + r.params,
                                    (torch.zeros_like(out),), allow_unused=True)
            del out
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    pool = torch.cuda.graph_pool_handle()
    # all forward graphs first, then the backward graphs in reverse order: the pool's liveness during capture
    # then mirrors the liveness during a training step (fwd 0..n-1, bwd n-1..0)
    prev = None
    for r in runners:
        if chain_input is not None and prev is not None and prev.static_out.shape == r.static_in[chain_input].shape \
                and prev.static_out.dtype == r.static_in[chain_input].dtype:
            # a fresh leaf over the previous pass's output storage (same strides): replaying pass k then pass k+1 hands
            # the queries over without a copy; any other tensor given at replay is copied into it as before
            r.static_in[chain_input] = prev.static_out.detach().requires_grad_(r.static_in[chain_input].requires_grad)
            r.chain_pos, r.chain_prev = chain_input, prev
        with torch.cuda.graph(r.fwd_graph, pool=pool, capture_error_mode=_CAPTURE_MODE):
            r.static_out = r.module(*r.static_in)
        prev = r
    for r in reversed(runners):
        r.static_grad_out = torch.zeros_like(r.static_out)
        targets = [r.static_in[j] for j in r.grad_in_idx] + r.params
        with torch.cuda.graph(r.bwd_graph, pool=pool, capture_error_mode=_CAPTURE_MODE):
            grads = torch.autograd.grad((r.static_out,), targets, (r.static_grad_out,), allow_unused=True)
            n_in = len(r.grad_in_idx)
            dst = [p.grad for p, g in zip(r.params, grads[n_in:]) if g is not None]
            src = [g for g in grads[n_in:] if g is not None]
            if dst:
                torch._foreach_add_(dst, src)
        r.static_grad_in = list(grads[:n_in])
        _GRAD_OUT_BUFFERS[(r.static_out.device.index, r.static_out.data_ptr())] = r.static_grad_out
        r.grad_ptrs = [p.grad.data_ptr() for p in r.params]
        r.data_ptrs = [p.data_ptr() for p in r.params]
    for r in runners:
        r.group = runners
        r.shared_pos = tuple(shared_inputs)
    torch.cuda.synchronize()
    return [GraphedPass(r) for r in runners]
