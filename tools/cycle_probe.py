#!/usr/bin/env python
"""Developer aid: which objects of a training step are only freed by the cyclic collector (reference cycles keep their
tensors alive until a generation-2 collection: allocator growth in long runs).  Usage (GPU box): python tools/cycle_probe.py"""
import gc
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

args = bench.parse(["--no-cpu-baseline", "--rotate", "0"])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
step = bench.make_mask3d_step(args, dev, 0, 1)
for _ in range(4):
    step(1)
torch.cuda.synchronize()
gc.collect()
gc.disable()
m0 = torch.cuda.memory_allocated()
for _ in range(3):
    step(1)
torch.cuda.synchronize()
m1 = torch.cuda.memory_allocated()
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
m2 = torch.cuda.memory_allocated()
print(f"allocated before {m0/2**20:.1f} MB, after 3 steps without collection {m1/2**20:.1f} MB, unreachable objects {n}")
cnt = Counter(type(o).__name__ for o in gc.garbage)
print(cnt.most_common(25))
tb = 0
for o in gc.garbage:
    if isinstance(o, torch.Tensor) and o.is_cuda:
        tb += o.numel() * o.element_size()
print(f"device tensors in cycles: {tb/2**20:.1f} MB")
seen = Counter()
for o in gc.garbage:
    t = type(o).__name__
    if t == "function":
        seen[f"function {o.__qualname__}"] += 1
    elif t == "cell":
        try:
            seen[f"cell -> {type(o.cell_contents).__name__}"] += 1
        except ValueError:
            seen["cell (empty)"] += 1
    elif t not in ("dict", "list", "tuple", "Tensor", "set"):
        seen[f"{t} {getattr(o, '__qualname__', '')}"] += 1
for k, v in seen.most_common(50):
    print(f"{v:6d}  {k}")
big = sorted((o for o in gc.garbage if isinstance(o, torch.Tensor) and o.is_cuda), key=lambda o: -o.numel() * o.element_size())[:12]
for o in big:
    print("tensor", tuple(o.shape), o.dtype, f"{o.numel()*o.element_size()/2**20:.1f} MB", "grad_fn" if o.grad_fn is not None else "")

# a cycle through a CoordinateManager: walk referents inside the garbage set until we come back
ids = {id(o): o for o in gc.garbage}
def desc(o):
    t = type(o).__name__
    if isinstance(o, torch.Tensor):
        return f"Tensor{tuple(o.shape)}"
    if t == "dict":
        return "dict{" + ",".join(str(k)[:18] for k in list(o)[:6]) + "}"
    if t in ("tuple", "list"):
        return f"{t}[{len(o)}]"
    if t == "function":
        return f"function {o.__qualname__}"
    if t == "cell":
        return "cell"
    return t
def find_cycle(start):
    stack = [(start, [start])]
    seen = set()
    while stack:
        o, path = stack.pop()
        for r in gc.get_referents(o):
            if id(r) not in ids:
                continue
            if r is start and len(path) > 1:
                return path
            if id(r) in seen:
                continue
            seen.add(id(r))
            stack.append((r, path + [r]))
    return None
shown = 0
for o in gc.garbage:
    if type(o).__name__ in ("CoordinateManager", "SparseTensor", "KMapRef", "SegmentCSR") and shown < 6:
        c = find_cycle(o)
        print("cycle from", type(o).__name__, ":", " -> ".join(desc(x) for x in c) if c else None)
        shown += 1
step.close()
