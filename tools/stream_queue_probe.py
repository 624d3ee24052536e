#!/usr/bin/env python
"""Which of torch's pool streams share a hardware queue with the default (null) stream, and with each other?
Usage (GPU box): python tools/stream_queue_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unscene3d_amd import streams  # noqa: E402

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
null = torch.cuda.default_stream(0)
normal = [torch.cuda.Stream(device=dev) for _ in range(9)]
high = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(5)]
print("ratio = time(two streams) / time(one): ~1 beside each other, ~2 one hardware queue")
print("normal pool stream i vs the null stream: ", [round(streams.overlap_ratio(null, s), 2) for s in normal])
print("high-priority pool stream i vs the null: ", [round(streams.overlap_ratio(null, s), 2) for s in high])
print("normal 0 vs normal i:                     ", [round(streams.overlap_ratio(normal[0], s), 2) for s in normal[1:]])
print("high 0 vs high i:                         ", [round(streams.overlap_ratio(high[0], s), 2) for s in high[1:]])
print("normal 0 vs high i:                       ", [round(streams.overlap_ratio(normal[0], s), 2) for s in high])
a = streams.pick(dev, "prefetch")
b = streams.pick(dev, "keys")
for r in streams.REPORT:
    print(r)
