"""Optimizer step in isolation: FlatAdamW (one kernel over flat buffers) against torch.optim.AdamW(fused=True) on a
parameter set shaped like the model's (ms per call incl. host issue)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unscene3d_amd.optim import FlatAdamW
dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(27, 96, 96)] * 120 + [(96,)] * 240 + [(27, 256, 256)] * 12 + [(128, 128)] * 60
def mk():
    return [torch.nn.Parameter(torch.randn(s, device=dev) * 0.01) for s in shapes]
pa, pb = mk(), mk()
oa = FlatAdamW(pa, lr=1e-4)
ob = torch.optim.AdamW(pb, lr=1e-4, fused=True)
for p in pb: p.grad = torch.randn_like(p)
oa.flat_grad.normal_()
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
n = sum(p.numel() for p in pa)
print("params", n / 1e6, "M")
print("flat step ms", t(oa.step), " torch fused step ms", t(ob.step))
print("flat zero_grad ms", t(lambda: oa.zero_grad()), " torch zero_grad ms", t(lambda: ob.zero_grad(set_to_none=False)))
