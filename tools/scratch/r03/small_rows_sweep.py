"""fwd+bwd device time (graph replay) of a 128->128 linear layer with a positional add, few-row kernels (SMALL_ROWS
raised) vs the many-row path, at the decoder's key counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
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
E = 128
for rows in (200, 800, 3200, 12800):
    res = []
    for small in (1024, 1 << 20):
        ops.SMALL_ROWS = small
        xq = torch.randn(100, 1, E, device=dev, requires_grad=True); pq = torch.randn(100, 1, E, device=dev, requires_grad=True)
        xk = torch.randn(rows, 1, E, device=dev, requires_grad=True); pk = torch.randn(rows, 1, E, device=dev)
        W = torch.randn(3 * E, E, device=dev, requires_grad=True); b = torch.randn(3 * E, device=dev, requires_grad=True)
        W.grad = torch.zeros_like(W); b.grad = torch.zeros_like(b)
        dq = torch.randn(100, 1, E, device=dev); dk = torch.randn(rows, 1, E, device=dev)
        def run():
            q, k, v = ops.in_proj(xq, xk, xk, W, b, pos_q=pq, pos_k=pk)
            torch.autograd.backward([q, k, v], [dq, dk, dk])
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            run()
            with torch.cuda.graph(g): run()
        torch.cuda.synchronize()
        res.append(t(g.replay))
    print(f"keys {rows:6d}: in_proj fwd+bwd  many-row path {res[0]:7.1f} us   few-row kernels {res[1]:7.1f} us")
