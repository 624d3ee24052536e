import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from unscene3d_amd.config import apply_overrides, default_config
from unscene3d_amd.datasets.synthetic import SyntheticFreeMaskDataset
from unscene3d_amd.datasets.utils import FreeMaskVoxelizeCollate
from unscene3d_amd.trainer.trainer import InstanceSegmentation
from test_gpu_parity import _PermSource
device = torch.device("cuda:0")
cfg = apply_overrides(default_config(), ["general.num_targets=3", "model.sample_sizes=[200,800,3200,12800,51200]"])
ds = SyntheticFreeMaskDataset(n_scenes=1, target_voxels=12000, seed=3300)
collate = FreeMaskVoxelizeCollate(ignore_label=255, voxel_size=0.02, mode="train", device=str(device))
torch.manual_seed(3)
eager = InstanceSegmentation(cfg).to(device).train()
graphed = InstanceSegmentation(cfg).to(device).train()
graphed.load_state_dict(eager.state_dict())
graphed.model.enable_decoder_graphs(batch_size=1, device=device)
res = []
for module in (eager, graphed):
    module.model.randperm = _PermSource()
    total, weighted = module.training_step(collate([ds[0]]))
    total.backward()
    res.append((float(total), {n: p.grad.clone() for n, p in module.named_parameters() if p.grad is not None}))
print("loss", res[0][0], res[1][0])
rows = []
for n in res[0][1]:
    a, b = res[0][1][n], res[1][1].get(n)
    if b is None:
        print("missing in graphed:", n); continue
    d = float((a - b).abs().max()); s = float(a.abs().max())
    rows.append((d / (s + 1e-12), d, s, n))
rows.sort(reverse=True)
for r in rows[:25]:
    print("%.3e  maxdiff %.3e  max %.3e  %s" % r)

# finite differences on the eager module for the worst bias entries
batch = collate([ds[0]])
def loss_of(module):
    module.model.randperm = _PermSource()
    with torch.no_grad():
        total, _ = module.training_step(batch)
    return float(total)
for name in ("model.lin_squeeze.0.3.bias", "model.cross_attention.0.3.multihead_attn.in_proj_bias", "model.lin_squeeze.0.2.bias"):
    a, b = res[0][1][name], res[1][1][name]
    idx = int((a - b).abs().argmax())
    p = dict(eager.named_parameters())[name]
    fds = []
    for eps in (3e-2, 1e-2):
        with torch.no_grad():
            p[idx] += eps; lp = loss_of(eager); p[idx] -= 2 * eps; lm = loss_of(eager); p[idx] += eps
        fds.append((lp - lm) / (2 * eps))
    print(name, idx, "eager", float(a[idx]), "graphed", float(b[idx]), "fd", fds)
    print("   eager[:6]", a[:6].tolist()); print("   graph[:6]", b[:6].tolist())
