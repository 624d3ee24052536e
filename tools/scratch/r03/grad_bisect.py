"""Where does the device gradient leave the f64 oracle's?  Gradients wrt the backbone's level outputs (aux[0..3] =
s16, s8, s4, s2 and the stride-1 features), device vs oracle f32 vs oracle f64, seed 3100, spatial_sort=5."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_step_parity as T
import oracle.mask3d_ref as OM
from unscene3d_amd.models.criterion import SetCriterion
dev = torch.device("cuda:0")
ss = 5 if len(sys.argv) < 2 else (int(sys.argv[1]) or False)
cfg, batch, collate, module = T._setup(dev, ss)
data, target, names = collate(batch)
# device: hook the level outputs
bb = module.model.backbone
orig = bb.forward
grabs = {}
def fwd(x):
    out, levels = orig(x)
    for i, lv in enumerate(list(levels) + [out]):
        lv.F.register_hook(lambda g, i=i: grabs.__setitem__(i, g.detach().cpu().double()))
    return out, levels
bb.forward = fwd
module.model.randperm = T.PermSource()
total, _ = module.training_step((data, target, names))
total.backward()
def oracle(dt):
    sd = T._leaves(module, dt)
    coords4 = data.coordinates.cpu().numpy(); feats = data.features.cpu(); p2s = [t["point2segment"].cpu() for t in target]
    out = OM.mask3d_forward(sd, cfg, coords4, feats[:, :3], feats[:, 3:], p2s, T.PermSource(), dtype=dt, keep_graph=True)
    lv = list(out["backbone_levels"])
    for t in lv: t.retain_grad()
    tgt_cpu = [{k: v.cpu() for k, v in t.items()} for t in target]
    crit = SetCriterion(num_classes=3, matcher=module.criterion.matcher, weight_dict=module.criterion.weight_dict, eos_coef=0.1,
                        losses=["labels", "masks"], num_points=-1, oversample_ratio=3.0, importance_sample_ratio=0.75, class_weights=-1)
    losses = crit(out, tgt_cpu, mask_type="segment_mask")
    wd = module.criterion.weight_dict
    sum(v * wd[k] for k, v in losses.items() if k in wd).backward()
    return [t.grad.double() for t in lv], sd
g32, sd32 = oracle(torch.float32)
g64, sd64 = oracle(torch.float64)
names_l = ["s16 (block4 out)", "s8 (block5 out)", "s4 (block6 out)", "s2 (block7 out)", "s1 (block8 out)"]
def re(a, b): return float((a - b).norm() / b.norm())
for i, n in enumerate(names_l):
    d = grabs[i]
    print(f"d loss / d {n:18s}: dev-vs-f64 {re(d, g64[i]):.2e}  cpu32-vs-f64 {re(g32[i], g64[i]):.2e}  dev-vs-cpu32 {re(d, g32[i]):.2e}   |g| {float(g64[i].norm()):.3e}", flush=True)
    # per-channel error profile of the worst level
    err = (d - g64[i]).norm(dim=0) / g64[i].norm(dim=0).clamp(min=1e-30)
    top = torch.topk(err, 5)
    print("      worst channels", [(int(c), f"{float(e):.1e}") for e, c in zip(top.values, top.indices)], " median", f"{float(err.median()):.1e}")
# BN statistics of the coarse levels: tiny-variance channels?
sd = module.model.state_dict()
for k in ("backbone.block4.5.norm2.bn", "backbone.block5.1.norm2.bn", "backbone.bntr4.bn", "backbone.bn4.bn"):
    rv = sd[k + ".running_var"].cpu()
    print(k, "running_var min", float(rv.min()), "median", float(rv.median()))
