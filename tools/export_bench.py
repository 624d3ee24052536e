"""Cost of the export post-processing at scene size: 150 k voxels, 240 k full-resolution
points, 100 queries, freemask settings (filter on, no DBSCAN) — device path vs the same host logic on torch CPU ops.

    python tools/export_bench.py [--reps 10]
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace as NS

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from unscene3d_amd import ops  # noqa: E402
from unscene3d_amd.trainer import postprocess as PP  # noqa: E402


def scene(seed, n_low=150_000, n_full=240_000, S=2500, Q=100):
    rng = np.random.default_rng(seed)
    p2s = rng.integers(0, S, n_low)
    inverse = np.concatenate([np.arange(n_low), rng.integers(0, n_low, n_full - n_low)])
    seg_full = p2s[inverse].copy()
    flip = rng.random(n_full) < 0.04
    seg_full[flip] = rng.integers(0, S, int(flip.sum()))
    obj = rng.integers(0, 40, S)
    masks = rng.normal(-4, 1, (S, Q)).astype(np.float32)
    for q in range(60):
        masks[:, q] = np.where(obj == q % 40, 4.0, -4.0)
    logits = rng.normal(0, 1, (1, Q, 3)).astype(np.float32)
    logits[0, :60, 1] += 3
    return p2s, inverse, seg_full, masks, logits


def run(dev, data, general):
    p2s, inverse, seg_full, masks, logits = data
    output = {"aux_outputs": [], "pred_logits": torch.from_numpy(logits).to(dev),
              "pred_masks": [torch.from_numpy(masks).to(dev)]}
    low = [{"point2segment": torch.from_numpy(p2s).to(dev)}]
    full = [{"point2segment": torch.from_numpy(seg_full).to(dev)}]
    inv = [torch.from_numpy(inverse).to(dev)]
    return lambda: PP.export_instances(output, low, full, inv, None, general, num_classes=3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    general = NS(use_dbscan=False, dbscan_eps=0.95, topk_per_image=100, filter_out_instances=True, scores_threshold=0.1,
                 iou_threshold=0.66)
    data = scene(0)
    fn = run(torch.device("cuda:0"), data, general)
    res = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / a.reps * 1e3

    # the same host logic on torch CPU ops (what the reference's CPU path costs)
    real = {k: getattr(ops, k) for k in ("gather_rows", "segment_csr", "segment_mean")}
    ops.gather_rows = lambda src, idx: src[idx]
    ops.segment_csr = lambda seg, S: ops.SegmentCSR(seg, None, None, S)

    def segment_mean(src, csr):
        out = torch.zeros((csr.S, src.shape[1])).index_add_(0, csr.seg, src)
        cnt = torch.zeros(csr.S).index_add_(0, csr.seg, torch.ones(src.shape[0]))
        return out / cnt.clamp(min=1)[:, None]

    ops.segment_mean = segment_mean
    cfn = run(torch.device("cpu"), data, general)
    t0 = time.perf_counter()
    cres = cfn()
    cpu_ms = (time.perf_counter() - t0) * 1e3
    for k, v in real.items():
        setattr(ops, k, v)
    same = bool(np.array_equal(res[0]["pred_masks"].cpu().numpy(), cres[0]["pred_masks"].numpy()))
    print(json.dumps({"voxels": 150_000, "points": 240_000, "queries": 100, "kept": int(res[0]["pred_masks"].shape[1]),
                      "device_ms_per_scene": gpu_ms, "torch_cpu_ms_per_scene": cpu_ms,
                      "cpu_threads": torch.get_num_threads(), "masks_equal": same}))


if __name__ == "__main__":
    main()
