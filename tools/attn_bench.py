"""Developer aid: fused masked cross attention vs the two-GEMM + softmax path (fwd+bwd device time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unscene3d_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
H, hd, L, B = 8, 16, 100, 1
E = H * hd
for S in (200, 800, 3200, 12800):
    q = torch.randn(L, B, E, device=dev, requires_grad=True); k = torch.randn(S, B, E, device=dev, requires_grad=True)
    v = torch.randn(S, B, E, device=dev, requires_grad=True); do = torch.randn(L, B, E, device=dev)
    mask = torch.rand(B, S, L, device=dev) > 0.5; mask[:, 0] = False
    def fused_f(): return ops.masked_cross_attention(q, k, v, mask, H)
    def fused(): fused_f().backward(do)
    def math_f():
        m = torch.zeros(B * H, L, S, device=dev).masked_fill_(mask.repeat_interleave(H, 0).permute(0, 2, 1), float("-inf"))
        qh = q.reshape(L, B * H, hd).transpose(0, 1); kh = k.reshape(S, B * H, hd).transpose(0, 1); vh = v.reshape(S, B * H, hd).transpose(0, 1)
        sc = torch.baddbmm(m, qh, kh.transpose(1, 2), alpha=0.25)
        return torch.bmm(torch.softmax(sc, -1), vh).transpose(0, 1).reshape(L, B, E)
    def math(): math_f().backward(do)
    print(f"S={S:6d}: fused fwd {t(fused_f):7.1f} us  fwd+bwd {t(fused):7.1f} us | math fwd {t(math_f):7.1f} us  fwd+bwd {t(math):7.1f} us")
