"""Which ATen operators (adds, copies, fills ...) one ungraphed training step issues: per phase (forward / loss /
backward / optimizer), per operator and operand shapes, with the python call site for the forward ones.

    python tools/op_census.py [--voxels 150000] [--top 80] > gpurun_out/op_census.txt

A TorchDispatchMode counts every operator that reaches the dispatcher (the autograd engine's threads inherit the mode);
the counts are what the captured decoder graphs replay as nodes and what the rest of the step issues through ATen."""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE)
import bench  # noqa: E402

SKIP = ("aten.view", "aten.detach", "aten.t.", "aten.permute", "aten.transpose", "aten.unsqueeze", "aten.squeeze",
        "aten.expand", "aten.select", "aten.slice", "aten.alias", "aten.as_strided", "aten._unsafe_view",
        "aten.reshape", "aten.unbind", "aten.split", "aten.empty", "aten.new_empty", "aten.lift_fresh",
        "aten.is_pinned", "aten.unflatten", "aten.narrow", "aten.chunk")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.phase = "forward"
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            shapes = tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor))[:3]
            on_dev = any(isinstance(a, torch.Tensor) and a.is_cuda for a in args)
            site = ""
            if True:     # custom Functions' backward runs Python too: the frame that issued the operator
                for fr in reversed(traceback.extract_stack(limit=14)):
                    if fr.filename.startswith(HERE) and "tools/op_census" not in fr.filename \
                            and not fr.filename.endswith("bench.py"):
                        site = f"{os.path.relpath(fr.filename, HERE)}:{fr.lineno}"
                        break
            if on_dev or not args:
                self.rows[(self.phase, name, shapes, site)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=150_000)
    ap.add_argument("--top", type=int, default=120)
    ap.add_argument("--graphs", action="store_true",
                    help="the step as bench.py runs it (decoder passes captured): what is listed is then the eager glue "
                         "BETWEEN the replays — the operators inside a captured pass never reach the dispatcher")
    a = ap.parse_args()
    args = bench.parse(["--voxels", str(a.voxels), "--no-prefetch", "--rotate", "0"] + ([] if a.graphs else ["--no-graphs"]))
    dev = torch.device("cuda:0")
    step = bench.make_mask3d_step(args, dev, 0, 1)
    for _ in range(3):
        step(1)
    torch.cuda.synchronize()
    census = Census()
    orig_backward = torch.Tensor.backward

    def backward(self, *aa, **kw):
        census.phase = "backward"
        try:
            return orig_backward(self, *aa, **kw)
        finally:
            census.phase = "optimizer"

    torch.Tensor.backward = backward
    with census:
        step(1)
        torch.cuda.synchronize()
    torch.Tensor.backward = orig_backward
    per_phase = collections.Counter()
    per_op = collections.Counter()
    for (phase, name, shapes, site), v in census.rows.items():
        per_phase[phase] += v
        per_op[(phase, name)] += v
    print("# device operators per phase (views / metadata ops excluded)")
    for k, v in per_phase.most_common():
        print(f"{v:6d}  {k}")
    print("# per phase and operator")
    for (phase, name), v in per_op.most_common(60):
        print(f"{v:6d}  {phase:9s} {name}")
    print("# per phase, operator, operand shapes, call site")
    for (phase, name, shapes, site), v in census.rows.most_common(a.top):
        print(f"{v:6d}  {phase:9s} {name:34s} {str(shapes):60s} {site}")


if __name__ == "__main__":
    main()
