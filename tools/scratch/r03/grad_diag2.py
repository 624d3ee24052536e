"""Device vs f64-oracle gradient error per backbone stage for several row orders / kernel families / seeds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_step_parity as T
from unscene3d_amd import ops, units
dev = torch.device("cuda:0")
STAGES = ["backbone.conv0p1s1", "backbone.block1", "backbone.block2", "backbone.block3", "backbone.block4", "backbone.convtr4p16s2",
          "backbone.block5", "backbone.convtr5p8s2", "backbone.bntr5", "backbone.block6", "backbone.block7", "backbone.block8", "cross_attention.0"]
def run(seed, ss):
    from unscene3d_amd.datasets.synthetic import SyntheticFreeMaskDataset
    cfg, batch, collate, module = T._setup(dev, ss)
    if seed != 3100:
        ds = SyntheticFreeMaskDataset(n_scenes=2, target_voxels=12000, seed=seed)
        batch = [ds[0], ds[1]]
    data, target, names = collate(batch)
    res = {}
    for dt in (torch.float32, torch.float64):
        sd = T._leaves(module, dt)
        tot, _ = T._oracle_step(module, cfg, sd, data, target, T.PermSource(), dt)
        tot.backward()
        res[dt] = sd
    for path, native in (("sorted", True), ("sorted", False), ("legacy", False)):
        ops.CONV_PATH = path; units.ENABLED = native
        for p in module.parameters(): p.grad = None
        module.model.randperm = T.PermSource()
        total, _ = module.training_step((data, target, names))
        total.backward()
        groups = {}
        for name, p in module.model.named_parameters():
            if name.startswith("backbone.final."): continue
            g64 = res[torch.float64][name].grad
            if float(g64.norm()) < 1e-12: continue
            grp = next((s for s in STAGES if name.startswith(s)), "other")
            a = groups.setdefault(grp, [0.0, 0.0, 0.0])
            a[0] += float((p.grad.double().cpu() - g64).square().sum()); a[1] += float((res[torch.float32][name].grad.double() - g64).square().sum()); a[2] += float(g64.square().sum())
        D = sum(a[0] for a in groups.values()); C = sum(a[1] for a in groups.values()); N = sum(a[2] for a in groups.values())
        print(f"== seed {seed} spatial_sort={ss} conv={path} native_units={native}: global dev {(D/N)**0.5:.2e} cpu32 {(C/N)**0.5:.2e}")
        print("   " + "  ".join(f"{g.split('.')[-1]}:{(groups[g][0]/groups[g][2])**0.5:.1e}/{(groups[g][1]/groups[g][2])**0.5:.1e}" for g in STAGES if g in groups), flush=True)
    ops.CONV_PATH = "sorted"; units.ENABLED = True
for seed, ss in ((3100, False), (3100, 5), (3100, 3), (3300, 5), (3300, False)):
    run(seed, ss)
