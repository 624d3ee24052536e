"""Per-launch timing of the conv kernels with HIP events on the launch stream
(bench.py's `roofline` object).  Inactive unless a
`capture()` block is open, so the timed region of bench.py is not perturbed."""
from __future__ import annotations

from contextlib import contextmanager

import torch

import os

_active = None
_pairs_cache = {}
SHAPES = os.environ.get("USC3D_PROF_SHAPES") == "1"   # developer aid: one aggregate per launch shape


def pick_nb(cout: int) -> int:   # mirrors csrc/spconv.hip pick_nb
    if cout <= 32:
        return 1
    if cout <= 64:
        return 2
    if cout % 128 == 0:
        return 4
    if cout % 96 == 0:
        return 3
    return 4


def table_pairs(nbr) -> int:
    key = (nbr.data_ptr(), tuple(nbr.shape))
    if key not in _pairs_cache:
        _pairs_cache[key] = int((nbr >= 0).sum().item())
    return _pairs_cache[key]


class Capture:
    def __init__(self):
        self.records = []

    @contextmanager
    def launch(self, kernel: str, flops: float, bytes_: float):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()   # torch's current stream == the stream handed to the C ABI
        yield
        e1.record()
        self.records.append((kernel, flops, bytes_, e0, e1))

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for k, fl, by, e0, e1 in self.records:
            a = agg.setdefault(k, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            a["launches"] += 1
            a["ms"] += e0.elapsed_time(e1)
            a["flops"] += fl
            a["bytes"] += by
        return agg

    def roofline(self, peak_tflops: float):
        agg = self.summary()
        if not agg:
            return None
        name, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
        achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
        total_ms = sum(v["ms"] for v in agg.values())
        return {
            "bound": "mfma", "kernel": name, "achieved": achieved, "peak": peak_tflops, "unit": "TFLOP/s",
            "frac": achieved / peak_tflops, "traffic": None, "launches": a["launches"],
            "avg_launch_us": 1e3 * a["ms"] / a["launches"],
            "algorithmic_gflop_per_launch": a["flops"] / a["launches"] / 1e9,
            "share_of_conv_time": a["ms"] / total_ms,
            "all_conv_kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                     "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else None}
                                 for k, v in agg.items()},
        }


@contextmanager
def capture():
    global _active
    cap = Capture()
    _active = cap
    try:
        yield cap
    finally:
        _active = None
        _pairs_cache.clear()


@contextmanager
def maybe(kernel_fn, flops_fn):
    """Used by ops: no-op unless a capture is open."""
    if _active is None:
        yield
    else:
        fl, by = flops_fn()
        with _active.launch(kernel_fn(), fl, by):
            yield
