import json, os, sys
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
args = bench.parse(["--gpus", "2", "--voxels", "40000", "--no-cpu-baseline", "--dist-backend", "gloo"])
step = bench.make_mask3d_step(args, dev, rank, world)
records = []
names = [n for n, _ in step.module.named_parameters()]
params = [p for _, p in step.module.named_parameters()]
def snap():
    torch.cuda.synchronize()
    records.append({n: float(p.grad.double().sum()) for n, p in zip(names, params) if p.grad is not None})
red = step.reducer
if red is not None:
    orig = red.finish
    def finish():
        r = orig(); snap(); return r
    red.finish = finish
else:
    orig_ar = dist.all_reduce
    def ar(t, *a, **k):
        r = orig_ar(t, *a, **k)
        return r
    import unscene3d_amd.optim as O
    orig_step = O.FlatAdamW.step
    def st(self, *a, **k):
        snap(); return orig_step(self, *a, **k)
    O.FlatAdamW.step = st
for _ in range(4):
    loss, _ = step(world)
torch.cuda.synchronize()
if rank == 0:
    json.dump({"loss": float(loss), "steps": records}, open(sys.argv[1], "w"))
dist.barrier()
