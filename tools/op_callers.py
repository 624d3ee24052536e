"""Developer aid: which Python call sites issue the small stock-PyTorch kernels of a training step (adds, copies,
fills)?  Runs a few steps under torch.profiler with stacks and groups the aten ops by their innermost frame inside
this repository."""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from unscene3d_amd import MinkowskiEngine as ME
from unscene3d_amd.config import apply_overrides, default_config
from unscene3d_amd.datasets.synthetic import SyntheticFreeMaskDataset
from unscene3d_amd.datasets.utils import FreeMaskVoxelizeCollate
from unscene3d_amd.trainer.trainer import InstanceSegmentation

ap = argparse.ArgumentParser()
ap.add_argument("--ops", default="aten::copy_,aten::add,aten::add_,aten::fill_,aten::zero_,aten::mul,aten::clone,aten::contiguous,aten::cat,aten::index,aten::to")
ap.add_argument("--top", type=int, default=45)
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = apply_overrides(default_config(), ["general.num_targets=3", "data.batch_size=1"])
torch.manual_seed(1234)
module = InstanceSegmentation(cfg).to(dev).train()
params = [p for n, p in module.named_parameters() if ".backbone.final." not in n]
opt = torch.optim.AdamW(params, lr=1e-4, fused=True)
sample = SyntheticFreeMaskDataset(n_scenes=1, target_voxels=150_000, seed=2000)[0]
sample = tuple(torch.from_numpy(np.ascontiguousarray(x)).to(dev) if isinstance(x, np.ndarray) and i in (0, 1, 2) else x
               for i, x in enumerate(sample))
module.model.enable_decoder_graphs(batch_size=1, device=dev)
collate = FreeMaskVoxelizeCollate(ignore_label=255, voxel_size=0.02, mode="train", device=str(dev))


def step():
    data, target, _ = collate([sample])
    feats = data.features
    raw = feats[:, -3:].contiguous()
    feats = feats[:, :-3].contiguous()
    x = ME.SparseTensor(coordinates=data.coordinates, features=feats, device=dev)
    out = module.forward(x, point2segment=[t["point2segment"] for t in target], raw_coordinates=raw)
    losses = module.criterion(out, target, mask_type=module.mask_type, coords=x.C)
    wd = module.criterion.weight_dict
    total = sum(v * wd[k] for k, v in losses.items() if k in wd)
    opt.zero_grad(set_to_none=False)
    total.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
N = 2
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                            with_stack=True, record_shapes=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
want = set(a.ops.split(","))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name not in want:
        continue
    site = None
    for fr in ev.stack:
        if root in fr and "tools/op_callers" not in fr:
            site = fr.replace(root + "/", "")
            break
    if site is None:   # no python frame recorded (autograd engine thread, or stacks unavailable): fall back to shapes
        site = "shapes " + str([tuple(s) for s in (ev.input_shapes or []) if s])[:90]
    k = (ev.name, site)
    agg[k][0] += 1
    agg[k][1] += ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print(f"{'op':14s} {'calls/step':>10s} {'dev us/step':>11s}  call site")
for (name, site), (cnt, us) in rows[:a.top]:
    print(f"{name:14s} {cnt / N:10.1f} {us / N:11.1f}  {site}")
