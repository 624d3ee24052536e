"""`felzenszwalb_cpp.segment_mesh` of the reference (pybind11 module built from utils/cpp_utils/segmentator.cpp:156-250;
called by pseudo_masks/datasets/scannet.py:176): graph-based over-segmentation of a coloured triangle mesh into
geometrically consistent segments + the directed segment adjacency.

    seg_indices, seg_connectivity = felzenszwalb_cpp.segment_mesh(vertices f32[V,3], faces i32[F,3], colors f32[V,3],
                                                                  kthr=0.005, segMinVerts=20)
    -> (i32[V] labels 0..S-1, i32[P,2] pairs (segment of a, segment of b) over the mesh edges, lexicographically sorted)

Normals, edge weights and the edge sort run on the MI355X (csrc/felz.hip, bit-equal weights); the two sequential
merge loops run on the host inside the library (usc_felz_merge_host).  Equal weights keep face order (stable sort)
where the reference's std::sort leaves their order to libstdc++: the partition is the same, the label NUMBERS (ranks of
union-find representatives) can differ — callers only use labels as ids."""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import check, lib, require_device


def edge_weights(vertices: torch.Tensor, faces: torch.Tensor, colors: torch.Tensor):
    """Device part: -> (edge_a i32[3F], edge_b i32[3F], weights f32[3F], normals f32[V,3]), edges in face order."""
    require_device()
    ops._chk(vertices, torch.float32, "vertices")
    ops._chk(colors, torch.float32, "colors")
    ops._chk(faces, torch.int32, "faces")
    V, F = vertices.shape[0], faces.shape[0]
    if F > 0:
        # the device kernels dereference the indices: reject a malformed mesh BEFORE the first launch (one small
        # reduction + read-back; the module's call ends in a host merge anyway)
        lo, hi = torch.aminmax(faces)
        if int(lo) < 0 or int(hi) >= V:
            raise RuntimeError(f"felzenszwalb_cpp: face index out of range [0, {V}) (min {int(lo)}, max {int(hi)})")
    dev = vertices.device
    st = ops._stream
    fn = torch.empty((F, 3), dtype=torch.float32, device=dev)
    check(lib.usc_felz_face_normals(vertices.data_ptr(), faces.data_ptr(), F, fn.data_ptr(), st()), "usc_felz_face_normals")
    csr = ops.segment_csr(faces.reshape(-1).to(torch.int64).contiguous(), V)     # stable: faces stay in face order
    normals = torch.empty((V, 3), dtype=torch.float32, device=dev)
    check(lib.usc_felz_vertex_normals(fn.data_ptr(), csr.order.data_ptr(), csr.seg_off.data_ptr(), V,
                                      normals.data_ptr(), st()), "usc_felz_vertex_normals")
    ea = torch.empty(3 * F, dtype=torch.int32, device=dev)
    eb = torch.empty(3 * F, dtype=torch.int32, device=dev)
    w = torch.empty(3 * F, dtype=torch.float32, device=dev)
    check(lib.usc_felz_edge_weights(vertices.data_ptr(), colors.data_ptr(), normals.data_ptr(), faces.data_ptr(), F,
                                    ea.data_ptr(), eb.data_ptr(), w.data_ptr(), st()), "usc_felz_edge_weights")
    return ea, eb, w, normals


def merge_host(edge_a: np.ndarray, edge_b: np.ndarray, weights: np.ndarray, n_vertices: int, kthr: float,
               seg_min_verts: int) -> np.ndarray:
    """The sequential merge loops over edges ALREADY sorted by weight (host arrays) -> representative per vertex."""
    a = np.ascontiguousarray(edge_a, np.int32)
    b = np.ascontiguousarray(edge_b, np.int32)
    w = np.ascontiguousarray(weights, np.float32)
    comps = np.empty(n_vertices, np.int32)
    check(lib.usc_felz_merge_host(a.ctypes.data, b.ctypes.data, w.ctypes.data, a.shape[0], int(n_vertices), float(kthr),
                                  int(seg_min_verts), comps.ctypes.data), "usc_felz_merge_host")
    return comps


def relabel(comps: np.ndarray, edge_a: np.ndarray, edge_b: np.ndarray):
    """The wrapper's output convention (segmentator.cpp:200-240): labels = rank of the representative, connectivity =
    sorted set of directed (segment of a, segment of b) pairs over all edges with different segments."""
    _, labels = np.unique(comps, return_inverse=True)
    labels = labels.reshape(-1).astype(np.int32)
    s1, s2 = labels[edge_a], labels[edge_b]
    keep = s1 != s2
    if not keep.any():
        return labels, np.zeros((0, 2), np.int32)
    return labels, np.unique(np.stack([s1[keep], s2[keep]], 1), axis=0).astype(np.int32)


def segment_mesh(vertices, faces, colors, kthr: float = 0.005, segMinVerts: int = 20, device="cuda"):
    dev = torch.device(device)
    to = lambda x, dt: (x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))).to(dev, dt).contiguous()
    v, c, f = to(vertices, torch.float32), to(colors, torch.float32), to(faces, torch.int32)
    ea, eb, w, _ = edge_weights(v, f, c)
    order = torch.sort(w, stable=True).indices            # equal weights keep face order
    host = torch.stack([ea[order].view(torch.float32), eb[order].view(torch.float32), w[order]]).cpu().numpy()   # one D2H
    sa, sb, sw = host[0].view(np.int32), host[1].view(np.int32), host[2]
    comps = merge_host(sa, sb, sw, v.shape[0], kthr, segMinVerts)
    return relabel(comps, sa, sb)
