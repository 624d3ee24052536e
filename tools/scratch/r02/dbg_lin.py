import sys, torch
sys.path.insert(0, '.')
from unscene3d_amd import ops
from unscene3d_amd.graphs import capture_passes
dev = torch.device("cuda:0")

class Lin(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.l = torch.nn.Linear(128, 128)
        self.mha = torch.nn.MultiheadAttention(128, 8)
    def forward(self, x, q):
        y = ops.linear(x, self.l.weight, self.l.bias)
        qq, kk, vv = ops.in_proj(q, y, y, self.mha.in_proj_weight, self.mha.in_proj_bias)
        return kk * 0.5 + vv + qq.sum() * 0.01

for M in (800, 3200, 12800):
    torch.manual_seed(0)
    m = Lin().to(dev)
    x = torch.randn(M, 1, 128, device=dev, requires_grad=True)
    q = torch.randn(100, 1, 128, device=dev, requires_grad=True)
    go = torch.randn(M, 1, 128, device=dev)
    # eager, autograd-returned grads
    out = m(x, q)
    params = [m.l.weight, m.l.bias, m.mha.in_proj_weight, m.mha.in_proj_bias]
    ge = torch.autograd.grad(out, [x, q] + params, go)
    (fn,) = capture_passes([m], [(x.detach().clone().requires_grad_(), q.detach().clone().requires_grad_())])
    for p in params:
        p.grad.zero_()
    for rep in range(3):
        xg, qg = x.detach().clone().requires_grad_(), q.detach().clone().requires_grad_()
        o = fn(xg, qg)
        o.backward(go)
    names = ["x", "q", "l.weight", "l.bias", "in_proj_weight", "in_proj_bias"]
    got = [xg.grad, qg.grad] + [p.grad / 3 for p in params]
    for n, a, b in zip(names, ge, got):
        print(M, n, "rel", float((a - b).abs().max() / (a.abs().max() + 1e-12)))
        if n == "in_proj_bias":
            for j in range(3):
                print("   slice", j, float((a[j*128:(j+1)*128] - b[j*128:(j+1)*128]).abs().max()), float(a[j*128:(j+1)*128].abs().max()))
