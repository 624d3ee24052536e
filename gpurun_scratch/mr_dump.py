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
for _ in range(4):
    loss, _ = step(world)
torch.cuda.synchronize()
if rank == 0:
    out = {"loss": float(loss)}
    for n, p in step.module.named_parameters():
        out[n] = [float(p.double().sum()), float(p.double().abs().sum())]
    json.dump(out, open(sys.argv[1], "w"))
dist.barrier()
