#!/usr/bin/env python
"""bench.py — training scenes/sec of UnScene3D's self-training hot path on MI355X.

`python bench.py --gpus N --steps K --warmup W`  (N>1: launched by torch.distributed.run,
one rank per GPU over RCCL).  One "step" = one pass of the hot path over one synthetic
ScanNet-shaped 150k-voxel scene per GPU: 2 cm voxelisation (hash unique) -> coordinate /
kernel maps -> Res16UNet34C forward -> backward -> (N>1: gradient all-reduce) -> AdamW.
Inputs (points, colours) are resident in HBM before the timed region.  Rank 0 prints ONE
JSON line (see DESIGN.md §Measurement for the roofline / cpu_baseline definitions).
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

VOXELS = 150_000
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--voxels", type=int, default=VOXELS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-voxels", type=int, default=15_000)
    return ap.parse_args()


def build_model(device):
    from unscene3d_amd.models.res16unet import Res16UNet34C

    torch.manual_seed(1234)
    cfg = SimpleNamespace(bn_momentum=0.02, conv1_kernel_size=3, dilations=[1, 1, 1, 1])
    return Res16UNet34C(3, 20, cfg, out_fpn=True).to(device).train()


def train_step(model, opt, xyz, colors, flat, world):
    from unscene3d_amd import MinkowskiEngine as ME
    from unscene3d_amd import ops

    # V1: voxelise on the device (reference: ME.utils.sparse_quantize in the collate, datasets/utils.py:403-408)
    c3, umap, _ = ME.utils.sparse_quantize(xyz, quantization_size=0.02, return_index=True, return_inverse=True,
                                           device=str(xyz.device))
    coords = torch.cat([torch.zeros((c3.shape[0], 1), dtype=torch.int32, device=xyz.device), c3], 1).contiguous()
    # collate-side row order: z-order cells keep neighbouring voxels' rows close in HBM/L2
    order = ops.spatial_order(coords)
    coords = ops.gather_rows_i32(coords, order)
    feats = ops.gather_rows(colors, umap[order])
    x = ME.SparseTensor(features=feats, coordinates=coords, device=xyz.device)   # V3
    out, fmaps = model(x)
    loss = out.F.square().mean()
    for f in fmaps[:-1]:
        loss = loss + f.F.square().mean()
    opt.zero_grad(set_to_none=False)
    loss.backward()
    if world > 1:
        dist.all_reduce(flat)          # RCCL over xGMI: one flat 151 MB buffer
        flat.div_(world)
    opt.step()
    return loss, coords.shape[0]


def cpu_baseline(sample_voxels):
    """Oracle (CPU restatement of the same fwd+bwd) on a bounded sample, rank 0 only."""
    import oracle.res16unet_ref as M
    from oracle import sparse_ref as R
    from unscene3d_amd.synthetic import make_scene

    sc = make_scene(2999, target_voxels=sample_voxels, tol=0.05)
    t0 = time.perf_counter()
    ec = R.voxel_floor(sc["xyz"], 0.02)
    eu, _ = R.sparse_quantize(ec)
    coords4, feats = R.sparse_collate([ec[eu]], [sc["colors"][eu]])
    feats = torch.from_numpy(feats)
    model = build_model("cpu") if False else None  # the device model cannot be built on CPU tensors
    from unscene3d_amd.models.res16unet import Res16UNet34C
    torch.manual_seed(1234)
    cfg = SimpleNamespace(bn_momentum=0.02, conv1_kernel_size=3, dilations=[1, 1, 1, 1])
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point)
          for k, v in Res16UNet34C(3, 20, cfg, out_fpn=True).state_dict().items()}
    pyr = M.Pyramid(coords4)
    out, levels = M.res16unet_forward(sd, pyr, feats, (2, 3, 4, 6, 2, 2, 2, 2))
    loss = out.square().mean()
    for f in levels[:-1]:
        loss = loss + f.square().mean()
    loss.backward()
    dt = time.perf_counter() - t0
    nv = coords4.shape[0]
    return {
        "value": (nv / VOXELS) / dt, "unit": "scenes/s (150k-voxel-scene equivalents)",
        "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"oracle Res16UNet34C fwd+bwd (voxelise+maps+conv/BN autograd, no optimizer) on one "
                  f"{nv}-voxel synthetic scene, {dt:.1f} s, scaled by voxels/150000",
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from unscene3d_amd import profiler
    from unscene3d_amd.ddp import flatten_grads
    from unscene3d_amd.synthetic import make_scene

    model = build_model(dev)
    params = [p for n, p in model.named_parameters() if not n.startswith("final.")]  # unused in forward
    flat = flatten_grads(params)
    opt = torch.optim.AdamW(params, lr=1e-4, fused=True)

    sc = make_scene(2000 + rank, target_voxels=args.voxels)
    xyz = torch.from_numpy(sc["xyz"]).to(dev)
    colors = torch.from_numpy(sc["colors"]).to(dev)

    for _ in range(args.warmup):
        loss, nvox = train_step(model, opt, xyz, colors, flat, world)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, nvox = train_step(model, opt, xyz, colors, flat, world)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # roofline of the dominant kernel: one extra instrumented step (HIP events around every conv launch on
    # the launch stream; not part of the timed region so that event overhead does not pollute `value`)
    roof = None
    if rank == 0:
        with profiler.capture() as prof:
            train_step(model, opt, xyz, colors, flat, 1)
            torch.cuda.synchronize()
        roof = prof.roofline(MFMA_F32_PEAK_TFLOPS)

    if rank == 0:
        line = {
            "metric": "training scenes/sec (Res16UNet34C+Mask3D, 150k voxels)",
            "value": world * args.steps / dt, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: Res16UNet34C backbone fwd+bwd+AdamW, one synthetic "
                                   f"ScanNet-shaped scene per GPU, {nvox} voxels @2cm (voxelise + coordinate/kernel "
                                   "maps rebuilt every step), random-init weights",
                       "voxels_per_scene": int(nvox), "global_batch": world, "parallelism": f"dp{world}",
                       "loss": float(loss)},
            "roofline": roof,
            "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline(args.cpu_sample_voxels),
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
