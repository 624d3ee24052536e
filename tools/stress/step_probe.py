#!/usr/bin/env python
"""Bit-reproducibility probe of the full self-training step (stress harness of tests/test_gpu_determinism.py).

Runs forward + criterion + backward of ONE fixed batch with FIXED weights and a fixed key-sampling stream `--iters`
times and compares the loss bits and every parameter gradient's bits with the first iteration.  `--load N` starts N
copies of itself on the same device first (what two ranks sharing the test GPU do to each other); `--graphs`
captures the decoder passes; `--prefetch` builds every batch on the side stream like bench.py.
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


class PermSource:
    def __init__(self):
        self.k = 0

    def __call__(self, n, device=None):
        g = torch.Generator().manual_seed(1000 + self.k)
        self.k += 1
        p = torch.randperm(n, generator=g)
        return p.to(device) if device is not None else p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--voxels", type=int, default=40000)
    ap.add_argument("--graphs", action="store_true")
    ap.add_argument("--prefetch", action="store_true")
    ap.add_argument("--load", type=int, default=0)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--spatial-sort", type=int, default=5)
    ap.add_argument("--seconds", type=float, default=0.0, help="child: run for this long")
    args = ap.parse_args()

    kids = []
    if args.load and not args.child:
        for _ in range(args.load):
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--voxels", str(args.voxels), "--iters", "100000",
                   "--seconds", "100000"] + (["--graphs"] if args.graphs else []) + (["--prefetch"] if args.prefetch else [])
            kids.append(subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))

    from unscene3d_amd.config import apply_overrides, default_config
    from unscene3d_amd.datasets.synthetic import SyntheticFreeMaskDataset
    from unscene3d_amd.datasets.utils import FreeMaskVoxelizeCollate
    from unscene3d_amd.ddp import flatten_grads
    from unscene3d_amd.trainer.trainer import InstanceSegmentation

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    cfg = apply_overrides(default_config(), ["general.num_targets=3"])
    torch.manual_seed(1234)
    module = InstanceSegmentation(cfg).to(dev).train()
    params = [p for n, p in module.named_parameters() if ".backbone.final." not in n]
    names = [n for n, p in module.named_parameters() if ".backbone.final." not in n]
    flat = flatten_grads(params)
    sample = SyntheticFreeMaskDataset(n_scenes=1, target_voxels=args.voxels, seed=2000)[0]
    sample = tuple(torch.from_numpy(np.ascontiguousarray(x)).to(dev) if isinstance(x, np.ndarray) and i in (0, 1, 2)
                   else x for i, x in enumerate(sample))
    if args.graphs:
        module.model.enable_decoder_graphs(batch_size=1, device=dev)
    collate = FreeMaskVoxelizeCollate(ignore_label=255, voxel_size=0.02, mode="train", device=str(dev),
                                      spatial_sort=args.spatial_sort)
    prefetch = None
    if args.prefetch:
        from unscene3d_amd.datasets.prefetch import ScenePrefetcher
        prefetch = ScenePrefetcher(collate, add_raw_coordinates=cfg.data.add_raw_coordinates, device=dev,
                                   precompute=module.model.precompute_geometry)
        prefetch.submit([sample])

    bounds, off = [], 0
    for p in params:
        bounds.append((off, off + p.numel()))
        off += p.numel()

    ref_flat, ref_loss = None, None
    bad = 0
    t_end = time.time() + args.seconds if args.child else None
    for it in range(args.iters):
        module.model.randperm = PermSource()
        batch = prefetch.take() if prefetch is not None else collate([sample])
        total, _ = module.training_step(batch)
        flat.zero_()
        total.backward()
        if prefetch is not None:
            prefetch.submit([sample])
        if args.child:
            if time.time() > t_end:
                break
            continue
        lbits = total.detach().view(torch.int32).item()
        if ref_flat is None:
            ref_flat, ref_loss = flat.clone(), lbits
            print(f"iter 0: loss {float(total):.9f}", flush=True)
            continue
        neq = flat.view(torch.int32) != ref_flat.view(torch.int32)
        if lbits != ref_loss or bool(neq.any()):
            bad += 1
            idx = torch.nonzero(neq).reshape(-1)
            which = []
            if idx.numel():
                pos = idx.cpu().numpy()
                starts = np.array([b[0] for b in bounds])
                pi = np.unique(np.searchsorted(starts, pos, side="right") - 1)
                which = [names[j] for j in pi]
            d = (flat - ref_flat).abs().max().item()
            print(f"iter {it}: loss bits {'same' if lbits == ref_loss else 'DIFF %.9f' % float(total)}, "
                  f"{idx.numel()} gradient words differ (max abs {d:.3e}) in {len(which)} params; first: {which[:6]} "
                  f"last: {which[-3:]}", flush=True)
    if not args.child:
        print(f"RESULT graphs={args.graphs} prefetch={args.prefetch} load={args.load}: {bad} of {args.iters - 1} "
              f"iterations differ from iteration 0", flush=True)
    for k in kids:
        k.kill()
    for k in kids:
        k.wait()


if __name__ == "__main__":
    main()
