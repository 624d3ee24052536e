#!/usr/bin/env python
"""Two-rank trajectory reproducibility probe (stress harness of tests/test_gpu_determinism.py).

Launched with torch.distributed.run --nproc-per-node 2 (ranks share device 0 on a one-GPU box: gloo; RCCL when the box
has two devices).  Every TRIAL resets weights, optimizer moments, LR schedule and the RNG to the same initial state and
runs `--steps` full training steps exactly like bench.py (graphs, prefetch, bucketed or single all-reduce); the loss
bits of every step and a checksum of the reduced gradient buffer before every optimizer step are compared with the
first trial.  `--load N`: N extra single-GPU training processes on the same device."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=40)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--voxels", type=int, default=40000)
    ap.add_argument("--load", type=int, default=0)
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true")
    a = ap.parse_args()
    import bench

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    if backend == "gloo":      # ranks share a device: keep the second streams off like bench.py does (see there)
        os.environ.setdefault("USC3D_WGRAD_LANE_MAX_ROWS", "0")
        os.environ.setdefault("USC3D_KV_SIDE_STREAM", "0")
    local = rank % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    kids = []
    if a.load and rank == 0:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
        for _ in range(a.load):
            kids.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "stress", "step_probe.py"), "--child", "--graphs",
                                          "--prefetch", "--iters", "1000000", "--seconds", "100000", "--voxels", "40000"],
                                         env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
    args = bench.parse(["--gpus", str(world), "--voxels", str(a.voxels), "--no-cpu-baseline", "--dist-backend", backend,
                        "--rotate", "0"]           # ONE scene replayed: every trial must see the same four batches
                       + (["--no-graphs"] if a.no_graphs else []) + (["--no-prefetch"] if a.no_prefetch else []))
    step = bench.make_mask3d_step(args, dev, rank, world)
    opt, module = step.opt, step.module
    p0 = opt.flat_param.clone()
    bufs0 = {n: b.clone() for n, b in module.named_buffers()}
    records = []
    cur = []

    def log_grads(o, x, k):       # the reduced, averaged gradients the optimizer is about to apply
        g = opt.flat_grad
        cur.append((float(g.double().sum()), float(g.double().abs().sum())))
    opt.register_step_pre_hook(log_grads)

    def reset():
        with torch.no_grad():
            opt.flat_param.copy_(p0)
            opt.exp_avg.zero_()
            opt.exp_avg_sq.zero_()
            opt.flat_grad.zero_()
            for n, b in module.named_buffers():
                b.copy_(bufs0[n])
        opt.steps = 0
        for g in opt.param_groups:
            g.pop("initial_lr", None)
        step.sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=module.config.optimizer.lr, total_steps=100000)
        step.set_sched(step.sched)
        torch.manual_seed(4321)

    bad = 0
    for t in range(a.trials):
        reset()
        cur.clear()
        losses = []
        for _ in range(a.steps):
            loss, _ = step(world)
            losses.append(loss.view(torch.int32).item())
        rec = (tuple(losses), tuple(cur))
        if not records:
            records.append(rec)
            if rank == 0:
                print(f"trial 0: loss {float(loss):.9f}", flush=True)
        elif rec != records[0]:
            bad += 1
            first = next(i for i in range(a.steps) if rec[0][i] != records[0][0][i] or rec[1][i] != records[0][1][i])
            print(f"rank {rank} trial {t}: differs from trial 0; first at step {first}: loss bits "
                  f"{rec[0][first]} vs {records[0][0][first]}, grad sums {rec[1][first]} vs {records[0][1][first]}; "
                  f"final loss {float(loss):.9f}", flush=True)
    print(f"RESULT rank {rank} world {world} backend {backend} graphs={not a.no_graphs} prefetch={not a.no_prefetch} "
          f"overlap={os.environ.get('USC3D_OVERLAP_ALLREDUCE', '1')} load={a.load}: {bad} of {a.trials - 1} trials differ",
          flush=True)
    for k in kids:
        k.kill()
    for k in kids:
        k.wait()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
