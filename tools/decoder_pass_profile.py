"""Developer aid: kernel list of ONE decoder pass (forward + backward, eager) per hlevel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from unscene3d_amd.config import default_config, apply_overrides, instantiate_model

dev = torch.device("cuda:0")
cfg = apply_overrides(default_config(), ["general.num_targets=3"])
torch.manual_seed(0)
model = instantiate_model(cfg).to(dev).train()
sizes = model.backbone.PLANES[-5:]
B, Q, d = 1, model.num_queries, model.mask_dim
for i, hlevel in enumerate(model.hlevels):
    K = model.sample_sizes[hlevel]
    ps = model._eager_pass(0, i)
    args = (torch.randn(B, Q, d, device=dev, requires_grad=True), torch.randn(Q, B, d, device=dev, requires_grad=True),
            torch.randn(B, K, sizes[hlevel], device=dev, requires_grad=True), torch.rand(B, K, Q, device=dev) > 0.5,
            torch.randn(B, K, d, device=dev))
    for _ in range(3):
        out = ps(*args); out.sum().backward()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        out = ps(*args)
        torch.cuda.synchronize()
        out.sum().backward()
        torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    tot = sum(e.device_time for e in ev)
    print(f"== hlevel {hlevel} K={K} C={sizes[hlevel]}: {len(ev)} kernels, {tot/1e3:.3f} ms device time")
    agg = {}
    for e in ev:
        a = agg.setdefault(e.name[:90], [0, 0.0]); a[0] += 1; a[1] += e.device_time
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"   {v[1]:8.1f} us {v[0]:3d}x  {k}")
