"""Developer aid: few-row linear kernels vs F.linear (forward + backward device time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unscene3d_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, k, n in [(100, 128, 128), (100, 128, 1024), (100, 1024, 128), (100, 128, 384)]:
    x = torch.randn(rows, k, device=dev, requires_grad=True); W = torch.randn(n, k, device=dev, requires_grad=True)
    b = torch.randn(n, device=dev, requires_grad=True); dy = torch.randn(rows, n, device=dev)
    def hip():
        y = ops.linear(x, W, b); y.backward(dy)
    def ref():
        y = torch.nn.functional.linear(x, W, b); y.backward(dy)
    g1 = torch.cuda.CUDAGraph(); g2 = torch.cuda.CUDAGraph()
    hip(); ref(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hip(); ref()
        with torch.cuda.graph(g1): hip()
        with torch.cuda.graph(g2): ref()
    torch.cuda.synchronize()
    print(f"rows {rows} in {k} out {n}: hip fwd+bwd {t(g1.replay):7.1f} us   torch {t(g2.replay):7.1f} us")
