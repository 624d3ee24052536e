"""Developer aid: the tile-form batch norm (usc_bn_tile_forward / _backward) against the launches it replaces
(slice reduction + statistics [+ finalisation] + apply; reduce + dx), per call, back to back on one stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unscene3d_amd._lib import check, lib

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()


def timeit(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


print(f"{'rows':>6} {'c':>4} {'G':>3} | fwd old  new | bwd old  new   (us per call)")
for n, c, G in [(507, 256, 27), (507, 128, 27), (2222, 128, 27), (2222, 256, 14), (9402, 128, 9), (9402, 64, 5), (9402, 128, 0),
                (2222, 256, 0), (40421, 32, 4), (40421, 96, 0)]:
    Gs = max(G, 1)
    parts = torch.randn(Gs, n, c, device=dev)
    y = torch.randn(n, c, device=dev)
    out = torch.empty(n, c, device=dev)
    res = torch.randn(n, c, device=dev)
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    stats = torch.empty(4, c, device=dev)
    red = torch.empty(2, c, device=dev)
    dgam, dbet = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    dout = torch.randn(n, c, device=dev)
    dy, dres = torch.empty(n, c, device=dev), torch.empty(n, c, device=dev)
    ws = torch.empty(int(max(lib.usc_colstats_ws_bytes(n, c), lib.usc_bn_tile_ws_bytes(c))), dtype=torch.uint8, device=dev)

    def fwd_old():
        if G > 0:
            check(lib.usc_group_reduce(p(parts), G, n, c, None, 0, p(y), st), "gr")
        check(lib.usc_bn_forward_stats(p(y), n, c, p(gamma), p(beta), 1e-5, 0.02, p(rm), p(rv), None, p(stats[0]), p(stats[1]),
                                       p(stats[2]), p(stats[3]), p(ws), ws.numel(), st), "st")
        check(lib.usc_bn_apply(p(y), p(stats[2]), p(stats[3]), p(res), 1, p(out), n, c, st), "ap")

    def fwd_new():
        check(lib.usc_bn_tile_forward(p(parts) if G > 0 else None, G, p(y), n, c, p(gamma), p(beta), 1e-5, 0.02, p(rm), p(rv),
                                      None, p(stats[0]), p(stats[1]), p(stats[2]), p(stats[3]), p(res), 1, p(out), p(ws),
                                      ws.numel(), st), "tf")

    def bwd_old():
        if G > 0:
            check(lib.usc_group_reduce(p(parts), G, n, c, None, 1, p(dout), st), "gr")
        check(lib.usc_bn_backward_reduce(p(y), p(dout), p(out), p(stats[0]), p(stats[1]), n, c, 1, 1, p(dgam), p(dbet),
                                         p(red[0]), p(red[1]), p(ws), ws.numel(), st), "br")
        check(lib.usc_bn_backward_dx(p(y), p(dout), p(out), p(stats[0]), p(stats[1]), p(gamma), p(red[0]), p(red[1]), p(dy),
                                     p(dres), n, c, st), "bd")

    def bwd_new():
        check(lib.usc_bn_tile_backward(p(parts) if G > 0 else None, G, 1, p(dout), p(y), p(out), p(stats[0]), p(stats[1]),
                                       p(gamma), n, c, 1, 1, p(dgam), p(dbet), p(dy), p(dres), p(ws), ws.numel(), st), "tb")

    fwd_old()
    if lib.usc_bn_tile_ok(n, c) or n > 12288:
        os.environ.setdefault("X", "1")
    t = [timeit(fwd_old), timeit(fwd_new) if lib.usc_bn_tile_ok(n, c) else float("nan"), timeit(bwd_old),
         timeit(bwd_new) if lib.usc_bn_tile_ok(n, c) else float("nan")]
    print(f"{n:6d} {c:4d} {G:3d} | {t[0]:7.1f} {t[1]:5.1f} | {t[2]:7.1f} {t[3]:5.1f}")
