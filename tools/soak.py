#!/usr/bin/env python
"""Soak run of the self-training step: N steps over R rotating scenes (119 k ... 178 k voxels) with the bench's step
object — what a long run must not show: allocator growth, a drifting step time, a non-finite loss, a late assignment
status.  Prints one JSON line.   Usage (GPU box): python tools/soak.py [--steps 400] [--rotate 8]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--rotate", type=int, default=8)
    a = ap.parse_args()
    args = bench.parse(["--no-cpu-baseline", "--rotate", str(a.rotate), "--rotate-spread", "0.2"])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    step = bench.make_mask3d_step(args, dev, 0, 1)
    for w in range(8):
        step(1)
        if w == 1 and os.environ.get("USC3D_STEADY", "1") == "1":
            from unscene3d_amd.trainer.trainer import prepare_steady_state
            print("steady state:", prepare_steady_state(dev), file=sys.stderr)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    marks, losses, mem = [], [], []
    for k in range(a.steps):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.append(ev)
        loss, _ = step(1)
        losses.append(loss)
        if k % 50 == 49:
            torch.cuda.synchronize()
            mem.append({"step": k + 1, "allocated_MB": round(torch.cuda.memory_allocated() / 2**20, 1),
                        "reserved_MB": round(torch.cuda.memory_reserved() / 2**20, 1)})
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append(ev)
    torch.cuda.synchronize()
    crit = step.module.criterion
    if hasattr(crit, "check_lsap_status"):
        crit.check_lsap_status(wait=True)
    per = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)])
    lv = torch.stack(losses).float().cpu().numpy()
    w = a.rotate * 5
    out = {"steps": a.steps, "rotating_scenes": a.rotate,
           "ms_per_step_first": float(per[:w].mean()), "ms_per_step_last": float(per[-w:].mean()),
           "ms_per_step_p50": float(np.median(per)), "ms_per_step_p99": float(np.quantile(per, 0.99)), "ms_per_step_max": float(per.max()),
           "loss_first": float(lv[:w].mean()), "loss_last": float(lv[-w:].mean()), "loss_all_finite": bool(np.isfinite(lv).all()),
           "peak_allocated_MB": round(torch.cuda.max_memory_allocated() / 2**20, 1), "memory": mem}
    step.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
