"""Developer aid: the bench's training step for a few dozen iterations on one scene — the loss must fall."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=40); ap.add_argument("--no-graphs", action="store_true")
a = ap.parse_args()
args = argparse.Namespace(voxels=150_000, no_graphs=a.no_graphs)
dev = torch.device("cuda", 0)
step = bench.make_mask3d_step(args, dev, 0, 1)
for it in range(a.iters):
    loss, n = step(1)
    if it % 5 == 0 or it == a.iters - 1:
        print(f"iter {it:3d} loss {float(loss):.4f}")
