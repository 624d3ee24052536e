"""Developer aid: wall time (with device sync) of the sections of one full self-training step."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--no-graphs", action="store_true"); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
args = argparse.Namespace(voxels=150_000, no_graphs=a.no_graphs)
dev = torch.device("cuda", 0)
from unscene3d_amd.config import apply_overrides, default_config
from unscene3d_amd.datasets.synthetic import SyntheticFreeMaskDataset
from unscene3d_amd.datasets.utils import FreeMaskVoxelizeCollate
from unscene3d_amd.trainer.trainer import InstanceSegmentation
from unscene3d_amd import MinkowskiEngine as ME

cfg = apply_overrides(default_config(), ["general.num_targets=3", "data.batch_size=1"])
torch.manual_seed(1234)
module = InstanceSegmentation(cfg).to(dev).train()
params = [p for n, p in module.named_parameters() if ".backbone.final." not in n]
opt = torch.optim.AdamW(params, lr=1e-4, fused=True)
sample = SyntheticFreeMaskDataset(n_scenes=1, target_voxels=150_000, seed=2000)[0]
sample = tuple(torch.from_numpy(np.ascontiguousarray(x)).to(dev) if isinstance(x, np.ndarray) and i in (0, 1, 2) else x for i, x in enumerate(sample))
if not a.no_graphs:
    module.model.enable_decoder_graphs(batch_size=1, device=dev)
collate = FreeMaskVoxelizeCollate(ignore_label=255, voxel_size=0.02, mode="train", device=str(dev), spatial_sort=False)

T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T.setdefault(name, []).append(time.perf_counter() - t0); return time.perf_counter()

bb = module.model.backbone
orig_bb = bb.forward
def timed_bb(*x, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = orig_bb(*x, **k); tick("  backbone fwd (inside model fwd)", t0); return r
bb.forward = timed_bb

for rep in range(a.reps + 2):
    if rep == 2: T.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    batch = collate([sample]); t0 = tick("collate (voxelise + targets)", t0)
    data, target, _ = batch
    feats = data.features; raw = feats[:, -3:].contiguous(); feats = feats[:, :-3].contiguous()
    x = ME.SparseTensor(coordinates=data.coordinates, features=feats, device=dev); t0 = tick("SparseTensor", t0)
    out = module.forward(x, point2segment=[t["point2segment"] for t in target], raw_coordinates=raw); t0 = tick("model fwd (backbone + decoder)", t0)
    losses = module.criterion(out, target, mask_type=module.mask_type, coords=x.C)
    wd = module.criterion.weight_dict
    total = sum(v * wd[k] for k, v in losses.items() if k in wd); t0 = tick("criterion fwd (match + losses)", t0)
    opt.zero_grad(set_to_none=False); t0 = tick("zero_grad", t0)
    total.backward(); t0 = tick("backward (all)", t0)
    opt.step(); t0 = tick("AdamW", t0)
tot = 0
for k, v in T.items():
    m = 1e3 * sum(v) / a.reps
    if not k.startswith("  "): tot += m
    print(f"{m:8.2f} ms  {k}")
print(f"{tot:8.2f} ms  total (sections are synchronised, so slightly above the free-running step)")
