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

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self._check_views()
        self.steps += 1
        b1, b2 = g["betas"]
        check(lib.usc_adamw_step(_ptr(self.flat_param), _ptr(self.flat_grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                                 self.flat_param.numel(), float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                 float(g["weight_decay"]), self.steps, _stream()), "usc_adamw_step")
        return loss

    def zero_grad(self, set_to_none: bool = False):
        if set_to_none:
            raise RuntimeError("FlatAdamW keeps the gradient buffer: zero_grad(set_to_none=False)")
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
