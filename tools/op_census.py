"""Which host-side ATen operators (adds, copies, fills ...) one ungraphed training step issues, and from where.

    python tools/op_census.py [--voxels 150000] [--top 60] > gpurun_out/op_census.txt

CPU-activity profile only (operator names + python call sites; no device tracing): the counts are what the captured
decoder graphs replay as nodes and what the backbone path issues through ATen each step."""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=150_000)
    ap.add_argument("--top", type=int, default=70)
    a = ap.parse_args()
    args = bench.parse(["--voxels", str(a.voxels), "--no-graphs", "--no-prefetch"])
    dev = torch.device("cuda:0")
    step = bench.make_mask3d_step(args, dev, 0, 1)
    for _ in range(3):
        step(1)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        step(1)
        torch.cuda.synchronize()
    want = ("aten::add", "aten::add_", "aten::copy_", "aten::fill_", "aten::zero_", "aten::mul", "aten::mul_",
            "aten::clone", "aten::contiguous", "aten::sum", "aten::index_select", "aten::cat", "aten::stack",
            "aten::where", "aten::masked_fill_", "aten::div", "aten::div_", "aten::sub", "aten::neg", "aten::relu",
            "aten::threshold_backward", "aten::_softmax", "aten::bmm", "aten::mm", "aten::addmm", "aten::index",
            "aten::index_put_", "aten::scatter_add_", "aten::gather", "aten::empty_like", "aten::zeros",
            "aten::zeros_like", "aten::sigmoid", "aten::exp", "aten::log", "aten::clamp", "aten::clamp_min")
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    by_site = collections.Counter()
    by_op = collections.Counter()
    for ev in prof.events():
        if ev.name not in want:
            continue
        site = "?"
        for fr in ev.stack:
            if here in fr and "tools/op_census" not in fr and "bench.py" not in fr:
                site = fr.replace(here + "/", "")
                break
        else:
            for fr in ev.stack:
                if "autograd" in fr or "Backward" in fr:
                    site = "autograd engine: " + fr[-80:]
                    break
        by_site[(ev.name, site)] += 1
        by_op[ev.name] += 1
    print("# operator totals (one step, graphs off)")
    for k, v in by_op.most_common():
        print(f"{v:6d}  {k}")
    print("# by call site")
    for (name, site), v in by_site.most_common(a.top):
        print(f"{v:6d}  {name:24s} {site}")


if __name__ == "__main__":
    main()
