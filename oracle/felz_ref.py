"""ctypes loader of oracle/libfelz_oracle.so (oracle/felz_oracle.cpp: C++ restatement of the reference's
utils/cpp_utils/segmentator.cpp) and, when it has been built, of the reference's own module under oracle/_ref/.
TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import glob
import importlib.util
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libfelz_oracle.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = C.CDLL(path)
        _LIB.felz_oracle_segment.restype = C.c_int
        _LIB.felz_oracle_segment.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                             C.c_int32, C.c_int32] + [C.c_void_p] * 5
    return _LIB


def canonical(comps, sorted_a, sorted_b):
    """The numpy wrapper's relabelling (segmentator.cpp:200-240): representatives -> 0..S-1 in ascending order, and the
    DIRECTED segment pairs (segment of a, segment of b) over all edges, lexicographically sorted (a std::map)."""
    uniq, labels = np.unique(comps, return_inverse=True)
    labels = labels.reshape(-1).astype(np.int32)
    s1, s2 = labels[sorted_a], labels[sorted_b]
    keep = s1 != s2
    pairs = np.unique(np.stack([s1[keep], s2[keep]], 1), axis=0) if keep.any() else np.zeros((0, 2), np.int32)
    return labels, pairs.astype(np.int32)


def segment_mesh(vertices, faces, colors, kthr=0.005, seg_min_verts=20, stable=False, details=False):
    """-> (labels i32[nv], connectivity i32[P,2]) like felzenszwalb_cpp.segment_mesh; `stable`: equal weights keep the
    input edge order (std::stable_sort) instead of libstdc++'s introsort order."""
    v = np.ascontiguousarray(vertices, np.float32)
    c = np.ascontiguousarray(colors, np.float32)
    f = np.ascontiguousarray(faces, np.int32)
    nv, nf = v.shape[0], f.shape[0]
    comps = np.empty(nv, np.int32)
    normals = np.empty((nv, 3), np.float32)
    weights = np.empty(3 * nf, np.float32)
    sa, sb = np.empty(3 * nf, np.int32), np.empty(3 * nf, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = _lib().felz_oracle_segment(p(v), p(c), p(f), nv, nf, float(kthr), int(seg_min_verts), int(stable), p(comps),
                                    p(normals), p(weights), p(sa), p(sb))
    assert rc == 0
    labels, pairs = canonical(comps, sa, sb)
    if details:
        return labels, pairs, {"normals": normals, "weights": weights, "comps": comps}
    return labels, pairs


def reference_module():
    """The reference's own felzenszwalb_cpp built by `make -C oracle ref` (None when it is not there)."""
    hits = glob.glob(os.path.join(_HERE, "_ref", "felzenszwalb_cpp*.so"))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location("felzenszwalb_cpp", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
