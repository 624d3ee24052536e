"""Per-scene cost of the mesh over-segmentation on a ScanNet-sized synthetic mesh: the device
path (normals, weights, stable sort on the MI355X; merge loops on the host) against the reference's own extension
module (oracle/_ref, built from /root/reference by `make -C oracle ref`) and against the C++ restatement."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import felz_ref as FR
from unscene3d_amd import felzenszwalb_cpp as FZ
from unscene3d_amd.synthetic import make_mesh

dev = torch.device("cuda:0")
v, f, c = make_mesh(21, side=400, n_regions=60)
dv, df, dc = (torch.from_numpy(x).to(dev) for x in (v, f, c))


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


out = {"vertices": int(v.shape[0]), "faces": int(f.shape[0]), "edges": int(3 * f.shape[0])}
out["device_path_ms"] = 1e3 * timed(lambda: FZ.segment_mesh(dv, df, dc, 0.005, 20, device=dev))
out["device_weights_only_ms"] = 1e3 * timed(lambda: FZ.edge_weights(dv, df, dc))
ea, eb, w, _ = FZ.edge_weights(dv, df, dc)
out["device_sort_ms"] = 1e3 * timed(lambda: torch.sort(w, stable=True))
order = torch.sort(w, stable=True).indices
sa, sb, sw = ea[order].cpu().numpy(), eb[order].cpu().numpy(), w[order].cpu().numpy()
t0 = time.perf_counter()
FZ.merge_host(sa, sb, sw, v.shape[0], 0.005, 20)
out["host_merge_ms"] = 1e3 * (time.perf_counter() - t0)
out["oracle_cpp_ms"] = 1e3 * timed(lambda: FR.segment_mesh(v, f, c, 0.005, 20, stable=False), n=3)
ref = FR.reference_module()
if ref is not None:
    out["reference_module_ms"] = 1e3 * timed(lambda: ref.segment_mesh(v, f, c, 0.005, 20), n=3)
    out["reference_module"] = "oracle/_ref (reference's segmentator.cpp, g++ -O2), one host core"
got = FZ.segment_mesh(dv, df, dc, 0.005, 20, device=dev)
out["segments"] = int(got[0].max()) + 1
print(json.dumps(out))
