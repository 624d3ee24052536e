import os, sys, warnings
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
import bench
for extra in (["--no-graphs"], []):
    args = bench.parse(["--voxels", "40000", "--no-cpu-baseline"] + extra)
    dev = torch.device("cuda:0")
    step = bench.make_mask3d_step(args, dev, 0, 1)
    losses = []
    for _ in range(3):
        loss, _ = step(1)
        losses.append(float(loss))
    bad = [n for n, p in step.module.named_parameters() if not torch.isfinite(p).all()]
    badg = [n for n, p in step.module.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(extra, "losses", losses, "non-finite params:", len(bad), bad[:8], "non-finite grads:", len(badg), badg[:8], flush=True)
    del step
    torch.cuda.empty_cache()
