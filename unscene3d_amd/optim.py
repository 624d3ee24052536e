"""AdamW over flat buffers (`usc_adamw_step`): the optimizer of the self-training step
(reference trainer/trainer.py:953-966: torch.optim.AdamW + OneCycleLR stepped per batch).

All parameters, gradients and both moments live in four flat fp32 buffers; every `p.data` / `p.grad` is a view.
One kernel per step reads and writes 28 B per parameter (torch's multi-tensor AdamW needs eight launches for the
300 tensors of this model and 2.4x the time), `zero_grad` is one memset, and the flat gradient buffer is the one the
data-parallel all-reduce works on (`ddp.flatten_grads` layout).  It is a `torch.optim.Optimizer`, so
`torch.optim.lr_scheduler.OneCycleLR` (which rewrites `lr` and `betas` every step) drives it unchanged."""
from __future__ import annotations

import torch

from ._lib import check, lib
from .ops import _ptr, _stream, require_device


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, flat_grad=None):
        params = list(params)
        if not params:
            raise ValueError("FlatAdamW: no parameters")
        if any(p.dtype != torch.float32 or not p.is_cuda for p in params):
            raise RuntimeError("FlatAdamW works on float32 HIP parameters")
        require_device()
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        dev = params[0].device
        total = sum(p.numel() for p in params)
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        if flat_grad is None:
            flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        elif flat_grad.numel() != total:
            raise RuntimeError("FlatAdamW: flat_grad does not match the parameters (use ddp.flatten_grads on the same list)")
        self.flat_grad = flat_grad
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                view = self.flat_param[off:off + n].view_as(p)
                view.copy_(p.data)
                p.data = view
                g = flat_grad[off:off + n].view_as(p)
                if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                    p.grad = g
                off += n
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.steps = 0
        self._params = params
        self._span = {}
        off = 0
        for p in params:
            self._span[id(p)] = (off, off + p.numel())
            off += p.numel()
        self._early_stream = None      # enable_early(): the stream the early ranges run on
        self._early_done = []          # [lo, hi) spans already stepped in the current step

    def _check_views(self):
        """Every p.data / p.grad must still be the view handed out in __init__ (module.to(), .float(),
        zero_grad(set_to_none=True) or a load_state_dict(assign=True) after construction would detach them, and the
        kernel would then update buffers nobody reads)."""
        lo, esz = self.flat_param.data_ptr(), self.flat_param.element_size()
        glo = self.flat_grad.data_ptr()
        off = 0
        for p in self._params:
            if p.data_ptr() != lo + off * esz:
                raise RuntimeError("FlatAdamW: a parameter no longer aliases the flat parameter buffer (moved or "
                                   "re-allocated after the optimizer was built); rebuild the optimizer")
            if p.grad is None or p.grad.data_ptr() != glo + off * esz:
                raise RuntimeError("FlatAdamW: a parameter's .grad no longer aliases the flat gradient buffer; keep "
                                   "gradients allocated (zero_grad(set_to_none=False))")
            off += p.numel()

    def _launch(self, lo, hi, step_no):
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        check(lib.usc_adamw_step(_ptr(self.flat_param[lo:hi]), _ptr(self.flat_grad[lo:hi]), _ptr(self.exp_avg[lo:hi]),
                                 _ptr(self.exp_avg_sq[lo:hi]), hi - lo, float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                 float(g["weight_decay"]), step_no, _stream()), "usc_adamw_step")

    def enable_early(self, stream):
        """Optimizer in the backward pass (single rank): parameters whose gradients are reported FINAL while backward is
        still being issued (ops.PARAMS_FINAL_HOOK — the step program reports a U-Net stage's parameters when the stage's
        kernels are queued) are stepped at once on `stream`, behind one event of the compute stream and one of the
        weight-gradient lane; step() covers the rest and lets the caller's stream wait for `stream`.  Same kernel, same
        per-element arithmetic: the trajectory keeps its bits.  What it takes off the end of the step: the 0.18 ms of the
        one 1.1 GB AdamW launch (39.6 M parameters x 28 B), of which the backbone's 95 % now run beside the backward pass.
        Not with a gradient reducer: a reduced bucket is what would have to trigger the range."""
        from . import ops
        self._early_stream = stream
        ops.PARAMS_FINAL_HOOK = self._on_final

    def disable_early(self):
        from . import ops
        if ops.PARAMS_FINAL_HOOK == self._on_final:
            ops.PARAMS_FINAL_HOOK = None
        self._early_stream = None

    @torch.no_grad()
    def _on_final(self, params):
        st = self._early_stream
        if st is None:
            return
        spans = sorted(self._span[id(p)] for p in params if id(p) in self._span)
        if not spans:
            return
        merged = [list(spans[0])]
        for lo, hi in spans[1:]:
            if lo <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], hi)
            else:
                merged.append([lo, hi])
        from . import units
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        st.wait_event(ev)
        lane_ev = units.lane_event(self.flat_param.device, release=False)
        if lane_ev is not None:
            st.wait_event(lane_ev)
        with torch.cuda.stream(st):
            for lo, hi in merged:
                self._launch(lo, hi, self.steps + 1)
        self._early_done.extend((lo, hi) for lo, hi in merged)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._check_views()
        self.steps += 1
        done, self._early_done = sorted(self._early_done), []
        if not done:
            self._launch(0, self.flat_param.numel(), self.steps)
            return loss
        torch.cuda.current_stream().wait_stream(self._early_stream)      # the early ranges' updates are in before the next forward
        pos = 0
        for lo, hi in done + [(self.flat_param.numel(), self.flat_param.numel())]:
            if lo > pos:
                self._launch(pos, lo, self.steps)
            pos = max(pos, hi)
        return loss

    def zero_grad(self, set_to_none: bool = False):
        if set_to_none:
            raise RuntimeError("FlatAdamW keeps the gradient buffer: zero_grad(set_to_none=False)")
        if self._early_done:       # a backward pass whose optimizer step never came (a skipped step): its early ranges HAVE been
            raise RuntimeError("FlatAdamW: early ranges of the previous backward pass were stepped but step() was not "
                               "called; call step() after every backward pass, or disable_early()")
        self.flat_grad.zero_()

    def state_dict(self):
        d = super().state_dict()
        d["flat"] = {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "steps": self.steps}
        return d

    def load_state_dict(self, state_dict):
        """Accepts this class's own state (key "flat") and the state of a torch.optim.AdamW over the same parameter
        list (the reference's checkpoints, trainer/trainer.py:953-966): its per-parameter exp_avg / exp_avg_sq / step
        are copied into the flat moment buffers.  The caller's dict is not modified."""
        state_dict = dict(state_dict)
        flat = state_dict.pop("flat", None)
        per_param = state_dict.get("state") or {}
        state_dict["state"] = {}
        super().load_state_dict(state_dict)
        if flat is not None:
            self.exp_avg.copy_(flat["exp_avg"])
            self.exp_avg_sq.copy_(flat["exp_avg_sq"])
            self.steps = int(flat["steps"])
        elif per_param:
            ids = [i for grp in state_dict["param_groups"] for i in grp["params"]]
            if len(ids) != len(self._params):
                raise RuntimeError("FlatAdamW.load_state_dict: the checkpoint covers a different parameter list")
            steps, off = set(), 0
            for i, p in zip(ids, self._params):
                st = per_param.get(i)
                n = p.numel()
                if st is not None:
                    if st["exp_avg"].numel() != n:
                        raise RuntimeError("FlatAdamW.load_state_dict: moment shape does not match its parameter")
                    self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.add(int(st["step"]))
                else:       # no state in the checkpoint (never stepped): fresh moments, not whatever was held before
                    self.exp_avg[off:off + n].zero_()
                    self.exp_avg_sq[off:off + n].zero_()
                off += n
            if len(steps) > 1:
                raise RuntimeError("FlatAdamW.load_state_dict: parameters with different step counts (one shared "
                                   "bias-correction step is kept)")
            self.steps = steps.pop() if steps else 0
