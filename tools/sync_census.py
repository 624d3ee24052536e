#!/usr/bin/env python
"""Developer aid: every host<->device synchronisation of one training step, by Python call site
(torch.cuda.set_sync_debug_mode("warn") turns each implicit sync — .item(), .cpu(), nonzero, a pageable copy —
into a warning that carries the line that issued it).
    python tools/sync_census.py [--voxels 150000] [--no-graphs] [--no-prefetch]"""
import argparse
import collections
import os
import sys
import warnings

import torch

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=150_000)
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true")
    a = ap.parse_args()
    argv = ["--voxels", str(a.voxels)] + (["--no-graphs"] if a.no_graphs else []) + (["--no-prefetch"] if a.no_prefetch else [])
    args = bench.parse(argv)
    dev = torch.device("cuda:0")
    step = bench.make_mask3d_step(args, dev, 0, 1)
    for _ in range(4):
        step(1)
    torch.cuda.synchronize()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
        step(1)
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    sites = collections.Counter()
    for w in rec:
        sites[(os.path.relpath(w.filename, HERE) if w.filename.startswith(HERE) else w.filename, w.lineno)] += 1
    print(f"# {sum(sites.values())} synchronising calls in one step")
    for (f, ln), v in sites.most_common():
        print(f"{v:5d}  {f}:{ln}")


if __name__ == "__main__":
    main()
