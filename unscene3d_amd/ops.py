"""Tensor-level wrappers over the C ABI (include/usc3d.h) + autograd Functions.

PyTorch is used only as plumbing here: device memory (torch.empty), the current
HIP stream, and autograd bookkeeping.  All arithmetic happens inside
libusc3d_hip.so.  Inputs must be contiguous tensors on a HIP device; anything
else raises RuntimeError (reference: CHECK_CONTIGUOUS / CHECK_CUDA,
utils/cuda_utils/cuda_utils.cpp:4-6, third_party/pointnet2/_ext_src/src/sampling.cpp:68).
"""
from __future__ import annotations

from dataclasses import dataclass

import ctypes
import os

import torch

from . import profiler as _prof
from ._lib import check, lib, require_device


def _stream():
    # raw handle of torch's current stream on the current device (torch.cuda.current_stream() builds a Stream
    # object through several Python layers: ~9 us per call, ~850 calls per training step)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _chk(t: torch.Tensor, dtype, name: str):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a HIP (cuda) tensor — unscene3d_amd has no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


def _ws(nbytes: int, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------ coordinates
def voxel_floor(xyz: torch.Tensor, voxel_size: float) -> torch.Tensor:
    """floor(xyz / voxel_size) in f64 -> i32[n,3] (reference datasets/utils.py:403)."""
    require_device()
    _chk(xyz, torch.float64, "xyz")
    out = torch.empty(xyz.shape, dtype=torch.int32, device=xyz.device)
    check(lib.usc_voxel_floor_f64(_ptr(xyz), xyz.shape[0], float(voxel_size), _ptr(out), _stream()),
          "usc_voxel_floor_f64")
    return out


@dataclass
class CoordMap:
    """One coordinate map: coordinates in row order + its HBM-resident hash table."""
    coords: torch.Tensor        # i32[n,4]  (b,x,y,z)
    table_keys: torch.Tensor    # i64[cap] (u64 bit pattern)
    table_vals: torch.Tensor    # i32[cap]
    tensor_stride: int

    @property
    def n(self) -> int:
        return self.coords.shape[0]

    @property
    def cap(self) -> int:
        return self.table_keys.shape[0]


def coordmap_build(coords: torch.Tensor, quant: int = 1, tensor_stride: int = 1):
    """-> (CoordMap of distinct coords, unique_idx i64[n_out], inverse i64[n]).

    First-occurrence unique == ME.utils.sparse_quantize(return_index, return_inverse)
    (reference datasets/utils.py:403-408)."""
    require_device()
    _chk(coords, torch.int32, "coords")
    if coords.dim() != 2 or coords.shape[1] != 4:
        raise RuntimeError("coords must be int32 [n,4] (b,x,y,z)")
    n = coords.shape[0]
    dev = coords.device
    cap = lib.usc_coordmap_capacity(n)
    keys = torch.empty(cap, dtype=torch.int64, device=dev)
    vals = torch.empty(cap, dtype=torch.int32, device=dev)
    unique_idx = torch.empty(n, dtype=torch.int64, device=dev)
    inverse = torch.empty(n, dtype=torch.int64, device=dev)
    out_coords = torch.empty((n, 4), dtype=torch.int32, device=dev)
    n_out = torch.zeros(1, dtype=torch.int64, device=dev)
    wsb = lib.usc_coordmap_ws_bytes(n)
    ws = _ws(wsb, dev)
    check(lib.usc_coordmap_build(_ptr(coords), n, int(quant), _ptr(keys), _ptr(vals), cap, _ptr(unique_idx),
                                 _ptr(inverse), _ptr(out_coords), _ptr(n_out), _ptr(ws), ws.numel(), _stream()),
          "usc_coordmap_build")
    m = int(n_out.item())
    if m < 0:
        raise RuntimeError("usc_coordmap_build: coordinate outside the packable range (|c| < 2^17, batch < 1024)")
    cmap = CoordMap(out_coords[:m].contiguous() if m != n else out_coords, keys, vals, tensor_stride)
    return cmap, unique_idx[:m], inverse


def spatial_order(coords: torch.Tensor, shift: int = 3) -> torch.Tensor:
    """Row permutation (i64[n]) that makes the rows of every 2^shift-voxel cell contiguous, cells in z-order,
    batches kept in order; stable inside a cell.  An optional collate-side optimisation (see usc3d.h)."""
    require_device()
    _chk(coords, torch.int32, "coords")
    n = coords.shape[0]
    dev = coords.device
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=dev)
    lo = coords[:, 1:].amin(0).tolist()
    hi = coords[:, 1:].amax(0).tolist()
    nb = int(coords[:, 0].max().item()) + 1
    extent = max(h - l for h, l in zip(hi, lo)) >> shift
    bits = max(1, int(extent).bit_length())
    if bits > 10:
        raise RuntimeError("spatial_order: scene extent too large for 10-bit cells; raise `shift`")
    ids = torch.empty(n, dtype=torch.int64, device=dev)
    check(lib.usc_morton_cell_ids(_ptr(coords), n, shift, lo[0], lo[1], lo[2], bits, _ptr(ids), _stream()),
          "usc_morton_cell_ids")
    return segment_csr(ids, nb << (3 * bits)).order


def kernel_map_cube(cmap: CoordMap, ksize: int = 3) -> torch.Tensor:
    """Dense neighbour table i32[K, n] of a stride-1 HYPER_CUBE kernel on `cmap`."""
    K = ksize ** 3
    nbr = torch.empty((K, cmap.n), dtype=torch.int32, device=cmap.coords.device)
    check(lib.usc_kernel_map_cube(_ptr(cmap.coords), cmap.n, cmap.tensor_stride, ksize, _ptr(cmap.table_keys),
                                  _ptr(cmap.table_vals), cmap.cap, _ptr(nbr), _stream()), "usc_kernel_map_cube")
    return nbr


def kernel_map_down2(fine: CoordMap, parent: torch.Tensor, coarse: CoordMap):
    """Child table i32[8, n_coarse] and kidx u8[n_fine] of a k=2,s=2 kernel."""
    dev = fine.coords.device
    nbr2 = torch.full((8, coarse.n), -1, dtype=torch.int32, device=dev)
    kidx = torch.empty(fine.n, dtype=torch.uint8, device=dev)
    check(lib.usc_kernel_map_down2(_ptr(fine.coords), fine.n, fine.tensor_stride, _ptr(parent), _ptr(coarse.coords),
                                   coarse.n, _ptr(nbr2), _ptr(kidx), _stream()), "usc_kernel_map_down2")
    return nbr2, kidx


class Rulebook:
    """Per-offset (in,out) pair lists.  in_idx/out_idx are allocated at capacity K*n_out; only the first
    koff[K] entries are meaningful.  `P` (the pair count) is read back from the device lazily — the
    kernels take the device-side koff and an upper bound, so the hot path never synchronises on it."""

    def __init__(self, in_idx, out_idx, koff, capacity, P=None):
        self.in_idx, self.out_idx, self.koff, self.capacity = in_idx, out_idx, koff, capacity
        self._P = P

    @property
    def P(self) -> int:
        if self._P is None:
            self._P = int(self.koff[-1].item())
        return self._P


def rulebook_compact(nbr: torch.Tensor) -> Rulebook:
    _chk(nbr, torch.int32, "nbr")
    K, n_out = nbr.shape
    dev = nbr.device
    in_idx = torch.empty(K * n_out, dtype=torch.int32, device=dev)
    out_idx = torch.empty(K * n_out, dtype=torch.int32, device=dev)
    koff = torch.empty(K + 1, dtype=torch.int64, device=dev)
    ws = _ws(lib.usc_rulebook_ws_bytes(K, n_out), dev)
    check(lib.usc_rulebook_compact(_ptr(nbr), K, n_out, _ptr(in_idx), _ptr(out_idx), _ptr(koff), _ptr(ws),
                                   ws.numel(), _stream()), "usc_rulebook_compact")
    return Rulebook(in_idx, out_idx, koff, K * n_out)


# ------------------------------------------------------------------ convolution
def _conv_cost(P, n_in, n_out, K, cin, cout):
    """Algorithmic work of one sparse-conv launch: flops = 2*P*cin*cout,
    bytes = 4*(n_in*cin + n_out*cout + K*cin*cout) + 8*P."""
    return 2.0 * P * cin * cout, 4.0 * (n_in * cin + n_out * cout + K * cin * cout) + 8.0 * P


def _sorted_symbol(cout):
    """The mask-sorted kernel's template instance for this output width (plan_sorted, csrc/spconv_sorted.hip)."""
    cb = cout // 32
    nb = 4 if cb % 4 == 0 else 3 if cb % 3 == 0 else 2 if cb % 2 == 0 else 1
    return f"usc::gather_gemm_sorted_kernel<{nb}, {4 if nb == 4 else 8}>"


def _kernel_symbol(kind, n, cin, cout, K):
    """rocprof-style symbol of the kernel a launch will use (usc_spconv_plan)."""
    code = lib.usc_spconv_plan(kind, int(n), cin, cout, K)
    nb, aligned, compact = code & 0xFF, (code >> 8) & 1, (code >> 12) & 1
    tag = f" [n={int(n)} cin={cin} cout={cout} K={K} G={code >> 16}]" if _prof.SHAPES else ""
    if kind == 0 and (code >> 14) & 1:
        return "usc::stem_conv_kernel" + tag
    if kind == 0 and compact:
        return f"usc::gather_gemm_compact_kernel<{nb}>" + tag
    if kind == 2:
        if (code >> 13) & 1:
            return f"usc::wgrad_full_kernel<{code >> 16}, {nb}>" + tag
        return f"usc::wgrad_kernel<{nb}, {'true' if aligned else 'false'}>" + tag
    base = "usc::gather_gemm_aligned_kernel" if aligned else "usc::gather_gemm_kernel"
    return f"{base}<{nb}, {'true' if kind == 1 else 'false'}>" + tag


def weight_transpose(W: torch.Tensor, mirror: bool) -> torch.Tensor:
    _chk(W, torch.float32, "W")
    K, cin, cout = W.shape
    out = torch.empty((K, cout, cin), dtype=torch.float32, device=W.device)
    check(lib.usc_weight_transpose(_ptr(W), K, cin, cout, int(mirror), _ptr(out), _stream()), "usc_weight_transpose")
    return out


def rowsort(nbr: torch.Tensor):
    """(perm i32[n_out], tile_mask i32[ceil(n_out/32)]) of a neighbour table: output rows grouped by their
    neighbour bitmask (usc_rowsort_build).  Built once per table and kept on the tensor object, so every conv
    on the same kernel map — forward and, for stride-1 maps, dgrad — shares it."""
    cached = getattr(nbr, "_usc_rowsort", None)
    if cached is not None:
        return cached
    _chk(nbr, torch.int32, "nbr")
    K, n_out = nbr.shape
    dev = nbr.device
    perm = torch.empty(n_out, dtype=torch.int32, device=dev)
    tmask = torch.empty((n_out + 31) // 32, dtype=torch.int32, device=dev)
    ws = _ws(lib.usc_rowsort_ws_bytes(K, n_out), dev)
    check(lib.usc_rowsort_build(_ptr(nbr), K, n_out, _ptr(perm), _ptr(tmask), _ptr(ws), ws.numel(), _stream()),
          "usc_rowsort_build")
    nbr._usc_rowsort = (perm, tmask)
    return perm, tmask


# "sorted" (default): mask-sorted kernel for small/medium maps, tile-compacted kernel for large ones;
# "sorted-all" / "legacy": force one family (A/B comparisons, tests)
CONV_PATH = os.environ.get("USC3D_CONV", "sorted")


def _sorted_applies(nbr, K, cin, cout):
    """Mask-sorted kernel for every aligned table-form conv EXCEPT the shapes the tile-compacted kernel takes
    (>= 24 k rows, cin >= 64): measured on the bench scene, 96->96 channels — 148 k rows: compacted 0.51 ms vs
    sorted 0.67 ms; 40 k rows: 0.19 vs 0.28; 9.4 k rows 128->128: row-order 0.146 vs sorted 0.105; 2.2 k rows
    256->256: 0.131 vs 0.086."""
    if CONV_PATH == "legacy" or nbr is None or not (1 < K <= 32) or cin % 32 or cout % 32 or cin > 4096:
        return False
    if CONV_PATH == "sorted-all":
        return True
    compact = (lib.usc_spconv_plan(0, int(nbr.shape[1]), cin, cout, K) >> 12) & 1
    return not compact


def gather_gemm(feats, W, nbr, n_out, bias=None, out=None, accumulate=False, w_transposed=False):
    """out[o] = sum_k feats[nbr[k,o]] @ W[k] (+bias).  W f32[K,cin,cout]; nbr None -> identity.
    w_transposed: W is the FORWARD conv's [K, cout, cin] and W'[k][c][n] = W[K-1-k][n][c] is meant (input gradient
    of a stride-1 conv); the sorted and tile-compacted kernels fold that into their weight staging / packing, the
    row-order kernels get an explicit usc_weight_transpose pass."""
    _chk(feats, torch.float32, "feats")
    _chk(W, torch.float32, "W")
    if w_transposed:
        K, cout, cin = W.shape
        linear_form = nbr is None and K == 1 and cin % 32 == 0 and cout % 32 == 0     # folded into the row-order kernel
        if not (linear_form or _sorted_applies(nbr, K, cin, cout) or
                (nbr is not None and (lib.usc_spconv_plan(0, int(n_out), cin, cout, K) >> 12) & 1)):
            W = weight_transpose(W, mirror=K > 1)
            w_transposed = False
    K, cin, cout = (W.shape[0], W.shape[2], W.shape[1]) if w_transposed else W.shape
    if feats.shape[1] != cin:
        raise RuntimeError(f"gather_gemm: feats have {feats.shape[1]} channels, kernel expects {cin}")
    if nbr is not None:
        _chk(nbr, torch.int32, "nbr")
        if nbr.shape[0] != K or nbr.shape[1] != n_out:
            raise RuntimeError("gather_gemm: neighbour table shape mismatch")
    if out is None:
        out = torch.empty((n_out, cout), dtype=torch.float32, device=feats.device)
    if bias is not None:
        _chk(bias, torch.float32, "bias")
    if _sorted_applies(nbr, K, cin, cout):
        perm, tmask = rowsort(nbr)
        wsb = lib.usc_spconv_sorted_ws_bytes(n_out, cin, cout, K)
        ws = _ws(wsb, feats.device) if wsb > 0 else None
        # (named per template instance like every other kernel of the capture — the way rocprofv3 lists them: the merged
        #  family once overtook the tile-compacted kernel as "dominant" by 0.2 ms between two boxes)
        with _prof.maybe(lambda: _sorted_symbol(cout) + (f" [n={n_out} cin={cin} cout={cout} K={K}]" if _prof.SHAPES else ""),
                         lambda: _conv_cost(_prof.table_pairs(nbr), feats.shape[0], n_out, K, cin, cout)):
            check(lib.usc_spconv_sorted_gemm(_ptr(feats), feats.shape[0], cin, _ptr(W), K, cout, _ptr(nbr), _ptr(perm),
                                             _ptr(tmask), n_out, _ptr(bias), _ptr(out), int(accumulate),
                                             int(w_transposed), _ptr(ws), wsb, _stream()), "usc_spconv_sorted_gemm")
        return out
    wsb = lib.usc_spconv_gather_gemm_ws_bytes(n_out, cin, cout, K)
    ws = _ws(wsb, feats.device) if wsb > 0 else None
    with _prof.maybe(lambda: _kernel_symbol(0, n_out, cin, cout, K),
                     lambda: _conv_cost(_prof.table_pairs(nbr) if nbr is not None else n_out, feats.shape[0], n_out,
                                        K, cin, cout)):
        check(lib.usc_spconv_gather_gemm(_ptr(feats), feats.shape[0], cin, _ptr(W), K, cout, _ptr(nbr), n_out,
                                         _ptr(bias), _ptr(out), int(accumulate), int(w_transposed), _ptr(ws), wsb,
                                         _stream()), "usc_spconv_gather_gemm")
    return out


def pairs_gemm(feats, W, rows_in, rows_out, koff, P, n_out):
    """out[rows_out[p]] = feats[rows_in[p]] @ W[k(p)] — every out row written exactly once."""
    _chk(feats, torch.float32, "feats")
    _chk(W, torch.float32, "W")
    K, cin, cout = W.shape
    if feats.shape[1] != cin:
        raise RuntimeError("pairs_gemm: channel mismatch")
    out = torch.empty((n_out, cout), dtype=torch.float32, device=feats.device)
    with _prof.maybe(lambda: _kernel_symbol(1, int(P), cin, cout, K),
                     lambda: _conv_cost(int(P), feats.shape[0], n_out, K, cin, cout)):
        check(lib.usc_spconv_pairs_gemm(_ptr(feats), cin, _ptr(W), K, cout, _ptr(rows_in), _ptr(rows_out),
                                        _ptr(koff), int(P), _ptr(out), _stream()), "usc_spconv_pairs_gemm")
    return out


# Parameter gradients straight into `p.grad`: when a leaf parameter already has a gradient buffer (the trainer
# keeps them allocated: zero_grad(set_to_none=False), one flat buffer for the all-reduce), the backward kernels add
# into it and the Function returns None, instead of returning a fresh tensor that autograd's AccumulateGrad then
# adds with one more small kernel per parameter (~250 launches per step for the backbone).
GRAD_IN_PLACE = os.environ.get("USC3D_GRAD_IN_PLACE", "1") == "1"


# streams other than the compute stream that backward kernels of this package write parameter gradients on (the decoder's
# key-preparation stream, models/mask3d.py): whoever reads p.grad outside autograd's own stream bookkeeping — a gradient
# reducer starting a collective in the middle of backward — lets the current stream wait for them first
SIDE_STREAMS = []


def on_side_stream():
    if not SIDE_STREAMS:
        return False
    raw = _stream()
    return any(st.cuda_stream == raw for st in SIDE_STREAMS)


def join_side_streams():
    cur = torch.cuda.current_stream()
    for st in SIDE_STREAMS:
        if st.device == cur.device:
            cur.wait_stream(st)


GRAD_WRITTEN_HOOK = None   # callable(param), set by ddp.BucketedGradReducer: "the kernels that add into p.grad are queued"


def _grad_target(param):
    """p.grad when the backward kernels may add into it directly, else None (the gradient is then returned to
    autograd).  Never when somebody else listens for the gradient through autograd: tensor hooks, or
    post-accumulate hooks (torch DDP, FSDP, user hooks) unless they belong to this package's own reducer, which is
    told about in-place writes through GRAD_WRITTEN_HOOK."""
    if GRAD_IN_PLACE and isinstance(param, torch.nn.Parameter) and param.is_leaf and param.grad is not None \
            and param.grad.is_contiguous() and not param._backward_hooks \
            and (GRAD_WRITTEN_HOOK is not None or not getattr(param, "_post_accumulate_grad_hooks", None)):
        return param.grad
    return None


_JOIN_QUEUED_TASK = None        # id of the autograd graph task whose end-of-backward join has been queued


def _graph_task():
    """Id of the running backward pass (-1 outside one).  The queued-once marks of the end-of-backward joins are keyed
    on it: the engine drops its callbacks when a backward node raises, and a sticky flag would then keep every later
    pass from queueing its join (tests/test_gpu_program.py::test_joins_are_queued_again_after_a_failed_backward)."""
    return torch._C._current_graph_task_id()


def _join_after_backward():
    """Called by backward kernels that may be running on one of SIDE_STREAMS.  The engine makes the caller's stream wait
    for the streams of the AccumulateGrad nodes it ran, not for a stream on which a backward wrote p.grad in place: one
    end-of-backward callback lets the caller's stream wait for the side streams, so that `loss.backward();
    optimizer.step()` stays correct without the caller knowing about them."""
    global _JOIN_QUEUED_TASK
    if not SIDE_STREAMS or not on_side_stream():
        return
    task = _graph_task()
    if task < 0 or task == _JOIN_QUEUED_TASK:
        return
    _JOIN_QUEUED_TASK = task

    def done():
        global _JOIN_QUEUED_TASK
        _JOIN_QUEUED_TASK = None
        join_side_streams()
    torch.autograd.Variable._execution_engine.queue_callback(done)


PARAMS_FINAL_HOOK = None  # callable(list of params), set by optim.FlatAdamW.enable_early: "every kernel that adds into the
                          # gradients of THESE parameters this step has been queued (compute stream and lane) — nothing
                          # else will" — only callers that know it by construction report (program.backward: a trunk
                          # unit's parameters are written once per backward pass)


def _params_final(params):
    if PARAMS_FINAL_HOOK is not None and params:
        PARAMS_FINAL_HOOK(params)


def _grad_written(*params):
    _join_after_backward()
    if GRAD_WRITTEN_HOOK is not None:
        for p in params:
            if p is not None:
                GRAD_WRITTEN_HOOK(p)


STEM_KERNEL = os.environ.get("USC3D_STEM_KERNEL", "1") != "0"


def wgrad(a, b, K, a_idx=None, b_idx=None, koff=None, into=None, out=None, nbr=None):
    """dW[k] = sum_{p in list k} a[a_idx[p]]^T b[b_idx[p]] -> f32[K,cin,cout]; ADDED into `into`, or WRITTEN to `out`
    (a contiguous buffer of K*cin*cout floats), when given.
    nbr (optional): the forward conv's neighbour table i32[K, n_out] of a stride-1 conv whose pair lists these are;
    the stem (<= 4 input channels, 32 output channels) then takes the table form (usc_spconv_wgrad_table)."""
    _chk(a, torch.float32, "a")
    _chk(b, torch.float32, "b")
    cin, cout = a.shape[1], b.shape[1]
    dW = into if into is not None else (out if out is not None else
                                        torch.empty((K, cin, cout), dtype=torch.float32, device=a.device))
    if STEM_KERNEL and nbr is not None and cin <= 4 and cout == 32 and K <= 32 and nbr.shape == (K, b.shape[0]):
        wsb = lib.usc_spconv_wgrad_table_ws_bytes(K, cin, cout)
        ws = _ws(wsb, a.device)
        with _prof.maybe(lambda: "usc::stem_wgrad_kernel" + (f" [n={b.shape[0]} cin={cin} cout={cout} K={K}]" if _prof.SHAPES else ""),
                         lambda: _conv_cost(_prof.table_pairs(nbr), a.shape[0], b.shape[0], K, cin, cout)):
            check(lib.usc_spconv_wgrad_table(_ptr(a), cin, _ptr(b), cout, _ptr(nbr), K, b.shape[0], _ptr(dW),
                                             int(into is not None), _ptr(ws), ws.numel(), _stream()),
                  "usc_spconv_wgrad_table")
        return dW
    ws = _ws(lib.usc_spconv_wgrad_ws_bytes(K, cin, cout), a.device)
    n_rows = a.shape[0] if a_idx is None else int(a_idx.shape[0])
    # (profiling only) real pair count = koff[K]; n_rows is the capacity of the pair lists
    with _prof.maybe(lambda: _kernel_symbol(2, n_rows, cin, cout, K),
                     lambda: _conv_cost(n_rows if koff is None else int(koff[-1].item()), a.shape[0], b.shape[0], K,
                                        cin, cout)):
        check(lib.usc_spconv_wgrad(_ptr(a), cin, _ptr(b), cout, K, _ptr(a_idx), _ptr(b_idx), _ptr(koff), n_rows,
                                   _ptr(dW), int(into is not None), _ptr(ws), ws.numel(), _stream()), "usc_spconv_wgrad")
    return dW


def wgrad_group(a_list, b_list, into_list, K, a_idx, b_idx, koff):
    """R same-shape weight gradients on one kernel map in ONE launch (usc_spconv_wgrad_group):
    into_list[r][k] += sum_p a_list[r][a_idx[p]]^T b_list[r][b_idx[p]].  The buffers of `into_list` are added into."""
    import ctypes as C
    R = len(a_list)
    if not (R == len(b_list) == len(into_list) and 1 <= R <= lib.usc_spconv_wgrad_group_max()):
        raise RuntimeError("wgrad_group: between 1 and usc_spconv_wgrad_group_max() problems, three lists of equal length")
    cin, cout = a_list[0].shape[1], b_list[0].shape[1]
    for a, b, w in zip(a_list, b_list, into_list):
        _chk(a, torch.float32, "a")
        _chk(b, torch.float32, "b")
        _chk(w, torch.float32, "dW")
        if a.shape[1] != cin or b.shape[1] != cout or w.numel() != K * cin * cout:
            raise RuntimeError("wgrad_group: the problems of a group have one shape")
    n_rows = a_list[0].shape[0] if a_idx is None else int(a_idx.shape[0])
    pa = (C.c_void_p * R)(*[t.data_ptr() for t in a_list])
    pb = (C.c_void_p * R)(*[t.data_ptr() for t in b_list])
    pw = (C.c_void_p * R)(*[t.data_ptr() for t in into_list])
    check(lib.usc_spconv_wgrad_group(R, pa, pb, pw, cin, cout, K, _ptr(a_idx), _ptr(b_idx), _ptr(koff), n_rows, 1,
                                     _stream()), "usc_spconv_wgrad_group")
    return into_list


class _ConvSame(torch.autograd.Function):
    """k^3 stride-1 conv on one map (in map == out map) or 1x1 conv (nbr None)."""

    @staticmethod
    def forward(ctx, feats, W, bias, nbr, get_rulebook):
        feats = feats.contiguous()
        W3 = W if W.dim() == 3 else W[None]
        out = gather_gemm(feats, W3.contiguous(), nbr, feats.shape[0], bias=None if bias is None else bias.reshape(-1))
        ctx.save_for_backward(feats, W3)
        ctx.nbr, ctx.get_rulebook = nbr, get_rulebook
        ctx.w_dim, ctx.has_bias = W.dim(), bias is not None
        ctx.w_param = W
        return out

    @staticmethod
    def backward(ctx, dout):
        feats, W3 = ctx.saved_tensors
        dout = dout.contiguous()
        K = W3.shape[0]
        dfeats = dW = dbias = None
        if ctx.needs_input_grad[0]:
            if K > 1:
                dfeats = gather_gemm(dout, W3.contiguous(), ctx.nbr, feats.shape[0], w_transposed=True)
            else:
                dfeats = gather_gemm(dout, weight_transpose(W3.contiguous(), mirror=False), ctx.nbr, feats.shape[0])
        if ctx.needs_input_grad[1]:
            tgt = _grad_target(ctx.w_param)
            if ctx.nbr is None:
                dW = wgrad(feats, dout, 1, into=tgt)
            else:
                rb = ctx.get_rulebook()
                dW = wgrad(feats, dout, K, rb.in_idx, rb.out_idx, rb.koff, into=tgt, nbr=ctx.nbr)
            if tgt is not None:
                dW = None
                _grad_written(ctx.w_param)
            elif ctx.w_dim == 2:
                dW = dW[0]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            dbias = colsum(dout).reshape(1, -1)
        return dfeats, dW, dbias, None, None


class _ConvDown2(torch.autograd.Function):
    """k=2, s=2 conv: fine map -> coarse map via the child table nbr2[8, n_coarse]."""

    @staticmethod
    def forward(ctx, feats, W, nbr2, get_rulebook):
        feats = feats.contiguous()
        out = gather_gemm(feats, W.contiguous(), nbr2, nbr2.shape[1])
        ctx.save_for_backward(feats, W)
        ctx.nbr2, ctx.get_rulebook = nbr2, get_rulebook
        ctx.w_param = W
        return out

    @staticmethod
    def backward(ctx, dout):
        feats, W = ctx.saved_tensors
        dout = dout.contiguous()
        rb = ctx.get_rulebook()  # in_idx = fine row, out_idx = coarse row
        dfeats = dW = None
        if ctx.needs_input_grad[0]:
            Wt = weight_transpose(W.contiguous(), mirror=False)
            dfeats = pairs_gemm(dout, Wt, rb.out_idx, rb.in_idx, rb.koff, feats.shape[0], feats.shape[0])
        if ctx.needs_input_grad[1]:
            tgt = _grad_target(ctx.w_param)
            dW = wgrad(feats, dout, W.shape[0], rb.in_idx, rb.out_idx, rb.koff, into=tgt)
            if tgt is not None:
                dW = None
                _grad_written(ctx.w_param)
        return dfeats, dW, None, None


class _ConvTrUp2(torch.autograd.Function):
    """k=2, s=2 transposed conv: coarse map -> cached fine map (same child table)."""

    @staticmethod
    def forward(ctx, feats, W, nbr2, get_rulebook, n_fine):
        feats = feats.contiguous()
        rb = get_rulebook()
        out = pairs_gemm(feats, W.contiguous(), rb.out_idx, rb.in_idx, rb.koff, n_fine, n_fine)
        ctx.save_for_backward(feats, W)
        ctx.nbr2, ctx.rb = nbr2, rb
        ctx.w_param = W
        return out

    @staticmethod
    def backward(ctx, dout):
        feats, W = ctx.saved_tensors
        dout = dout.contiguous()
        rb = ctx.rb
        dfeats = dW = None
        if ctx.needs_input_grad[0]:
            Wt = weight_transpose(W.contiguous(), mirror=False)
            dfeats = gather_gemm(dout, Wt, ctx.nbr2, feats.shape[0])
        if ctx.needs_input_grad[1]:
            tgt = _grad_target(ctx.w_param)
            dW = wgrad(feats, dout, W.shape[0], rb.out_idx, rb.in_idx, rb.koff, into=tgt)
            if tgt is not None:
                dW = None
                _grad_written(ctx.w_param)
        return dfeats, dW, None, None, None


def conv_same(feats, W, bias, nbr, get_rulebook):
    return _ConvSame.apply(feats, W, bias, nbr, get_rulebook)


def conv_down2(feats, W, nbr2, get_rulebook):
    return _ConvDown2.apply(feats, W, nbr2, get_rulebook)


def conv_tr_up2(feats, W, nbr2, get_rulebook, n_fine):
    return _ConvTrUp2.apply(feats, W, nbr2, get_rulebook, n_fine)


# ------------------------------------------------------------------ batch norm / relu
def colstats(x, y=None):
    """-> (sum_i x[i,c], sum_i x[i,c]*y[i,c]) as f64[c] (y None -> x*x)."""
    _chk(x, torch.float32, "x")
    n, c = x.shape
    s1 = torch.empty(c, dtype=torch.float64, device=x.device)
    s2 = torch.empty(c, dtype=torch.float64, device=x.device)
    ws = _ws(lib.usc_colstats_ws_bytes(n, c), x.device)
    check(lib.usc_colstats(_ptr(x), _ptr(y), n, c, _ptr(s1), _ptr(s2), _ptr(ws), ws.numel(), _stream()),
          "usc_colstats")
    return s1, s2


def colsum(x):
    return colstats(x)[0].to(torch.float32)


def bn_apply(x, scale, shift, residual=None, relu=False):
    _chk(x, torch.float32, "x")
    n, c = x.shape
    y = torch.empty_like(x)
    check(lib.usc_bn_apply(_ptr(x), _ptr(scale), _ptr(shift), _ptr(residual), int(relu), _ptr(y), n, c, _stream()),
          "usc_bn_apply")
    return y


class _BatchNormAct(torch.autograd.Function):
    """y = [relu](BN_train(x) [+ residual]) — MinkowskiBatchNorm (+MinkowskiReLU) (+`out += residual`)
    fused into one stats pass and one apply pass (reference models/modules/resnet_block.py:48-64)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, relu, eps, running_mean, running_var, momentum, training,
                num_batches_tracked=None):
        x = x.contiguous()
        n, c = x.shape
        dev = x.device
        if training:
            stats = torch.empty((4, c), dtype=torch.float32, device=dev)   # mean | invstd | scale | shift
            mean, invstd, scale, shift = stats[0], stats[1], stats[2], stats[3]
            ws = _ws(lib.usc_colstats_ws_bytes(n, c), dev)
            check(lib.usc_bn_forward_stats(_ptr(x), n, c, _ptr(gamma), _ptr(beta), float(eps), float(momentum),
                                           _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked),
                                           _ptr(mean), _ptr(invstd), _ptr(scale), _ptr(shift), _ptr(ws), ws.numel(),
                                           _stream()),
                  "usc_bn_forward_stats")
        else:
            mean = running_mean
            invstd = torch.rsqrt(running_var + eps)
            scale = (gamma * invstd).contiguous()
            shift = (beta - mean * scale).contiguous()
        res = None if residual is None else residual.contiguous()
        y = bn_apply(x, scale, shift, res, relu)
        ctx.save_for_backward(x, gamma, mean, invstd, y if relu else None)
        ctx.relu, ctx.has_res, ctx.training = relu, residual is not None, training
        ctx.gamma_param, ctx.beta_param = gamma, beta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, invstd, y = ctx.saved_tensors
        dy = dy.contiguous()
        n, c = x.shape
        dev = x.device
        red = torch.empty((4, c), dtype=torch.float32, device=dev)   # dgamma | dbeta | mean_g | mean_gx
        ws = _ws(lib.usc_colstats_ws_bytes(n, c), dev)
        tg, tb = _grad_target(ctx.gamma_param), _grad_target(ctx.beta_param)
        in_place = tg is not None and tb is not None
        dg, db = (tg, tb) if in_place else (red[0], red[1])
        check(lib.usc_bn_backward_reduce(_ptr(x), _ptr(dy), _ptr(y), _ptr(mean), _ptr(invstd), n, c,
                                         int(ctx.training), int(in_place), _ptr(dg), _ptr(db), _ptr(red[2]), _ptr(red[3]),
                                         _ptr(ws), ws.numel(), _stream()), "usc_bn_backward_reduce")
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        check(lib.usc_bn_backward_dx(_ptr(x), _ptr(dy), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(red[2]),
                                     _ptr(red[3]), _ptr(dx), _ptr(dres), n, c, _stream()), "usc_bn_backward_dx")
        if in_place:
            _grad_written(ctx.gamma_param, ctx.beta_param)
        return (dx, (None if in_place else red[0]), (None if in_place else red[1]), dres, None, None, None, None, None,
                None, None)


def batch_norm_act(x, gamma, beta, residual=None, relu=False, eps=1e-5, running_mean=None, running_var=None,
                   momentum=0.1, training=True, num_batches_tracked=None):
    """`num_batches_tracked` (i64[1], optional) is incremented inside the statistics launch of a training pass."""
    return _BatchNormAct.apply(x, gamma, beta, residual, relu, eps, running_mean, running_var, momentum, training,
                               num_batches_tracked)


class _ReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib.usc_relu_fwd(_ptr(x), _ptr(y), x.numel(), _stream()), "usc_relu_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(lib.usc_relu_bwd(_ptr(y), _ptr(dy), _ptr(dx), y.numel(), _stream()), "usc_relu_bwd")
        return dx


def relu(x):
    _chk(x, torch.float32, "x")
    return _ReLU.apply(x)


# ------------------------------------------------------------------ pooling / gather / segments
def avgpool_down2(feats, nbr2, row_of=None, threshold=False):
    """MinkowskiAvgPooling(kernel_size=2, stride=2) forward: mean over present children.
    row_of (i64[n_fine]): the fine rows are feats[row_of[i]] (feats may be a column slice of a wider table); a child
    table that already holds the ROWS OF `feats` to read (row_of folded into nbr2 once per batch) needs no row_of;
    threshold: return sigmoid(mean) < 0.5 as bool instead of the means."""
    if feats.dtype != torch.float32 or not feats.is_cuda or feats.dim() != 2 or feats.stride(1) != 1:
        raise RuntimeError("avgpool_down2: feats must be an f32 HIP matrix with unit column stride")
    # (a column slice of a wider table is fine either way: rows are addressed through the leading dimension)
    if row_of is not None:
        _chk(row_of, torch.int64, "row_of")
    nc, c = nbr2.shape[1], feats.shape[1]
    ld = feats.stride(0) if feats.shape[0] > 1 else c
    out = torch.empty((nc, c), dtype=torch.bool if threshold else torch.float32, device=feats.device)
    check(lib.usc_avgpool_down2_ex(_ptr(feats), c, ld, _ptr(row_of), _ptr(nbr2), nc,
                                   None if threshold else _ptr(out), _ptr(out) if threshold else None, _stream()),
          "usc_avgpool_down2_ex")
    return out


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, idx, out=None, unique=False):
        src = src.contiguous()
        idx = idx.contiguous()
        if out is None:
            out = torch.empty((idx.shape[0], src.shape[1]), dtype=torch.float32, device=src.device)
        elif tuple(out.shape) != (idx.shape[0], src.shape[1]) or out.dtype != torch.float32 or not out.is_contiguous() \
                or out.requires_grad:
            raise RuntimeError("gather_rows: `out` must be a contiguous f32 [len(idx), C] tensor that does not require grad")
        check(lib.usc_gather_rows(_ptr(src), src.shape[1], _ptr(idx), idx.shape[0], _ptr(out), _stream()),
              "usc_gather_rows")
        ctx.save_for_backward(idx)
        ctx.n_src, ctx.unique = src.shape[0], bool(unique)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dout = dout.contiguous()
        dsrc = torch.zeros((ctx.n_src, dout.shape[1]), dtype=torch.float32, device=dout.device)
        fn = lib.usc_scatter_rows_unique if ctx.unique else lib.usc_scatter_add_rows
        check(fn(_ptr(dout), dout.shape[1], _ptr(idx), idx.shape[0], _ptr(dsrc), _stream()), "usc_scatter_add_rows")
        return dsrc, None, None, None


def gather_rows(src, idx, out=None, unique=False):
    """out[j] = src[idx[j]]; `out` (optional): the caller's buffer (e.g. a HIP-graph input) instead of a new tensor.
    unique: the caller guarantees that idx holds no duplicates (a permutation / torch.randperm(n)[:k]); the backward
    pass then scatters with plain stores instead of float atomics."""
    _chk(src, torch.float32, "src")
    _chk(idx, torch.int64, "idx")
    return _GatherRows.apply(src, idx, out, unique)


class GradSink:
    """One gradient buffer for SEVERAL consumers of the same table (the three decoders' key samples of a backbone
    level): every consumer's backward accumulates its rows into `buf` (zero-filled by the first one to arrive) and hands
    autograd None; the last one hands over the buffer.  Replaces one zero fill + scatter per consumer and autograd's
    adds of the dense results (same per-row summation order: arrival order).  A consumer whose backward never runs would
    leave the others' gradients stranded: checked when the backward pass ends."""

    def __init__(self):
        self.buf, self.pending, self.check_queued = None, 0, False

    def _check(self):
        self.check_queued = False
        if self.pending != 0 and self.buf is not None:
            self.buf = None
            raise RuntimeError("GradSink: a consumer's backward did not run; the gradients accumulated for its table were "
                               "not delivered (call the consumers without a shared sink)")


class _SampleKeys(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, mask, pos, idx, n_scenes, K, n_valid, outs, unique, sink=None, valid_unique=False):
        # feats None: the mask rows only; mask None: the feature (+ positional) rows only (usc_sample_keys' partial calls)
        c = 0 if feats is None else feats.shape[1]
        q = 0 if mask is None else mask.shape[1]
        p = 0 if pos is None else pos.shape[1]
        dev = idx.device
        if outs is None:
            of = None if feats is None else torch.empty((n_scenes, K, c), dtype=torch.float32, device=dev)
            om = None if mask is None else torch.empty((n_scenes, K, q), dtype=torch.bool, device=dev)
            op = None if pos is None else torch.empty((n_scenes, K, p), dtype=torch.float32, device=dev)
        else:
            # fresh tensor objects over the caller's storage: the objects returned from here get autograd metadata
            of, om, op = (None if o is None else o.detach() for o in outs)
            for name, t, want, dt in (("features", of, (n_scenes, K, c), torch.float32), ("mask", om, (n_scenes, K, q), torch.bool),
                                      ("positions", op, (n_scenes, K, p), torch.float32)):
                if (t is None) != (want[2] == 0):
                    raise RuntimeError(f"sample_keys: an output buffer for the {name} is {'missing' if t is None else 'not wanted'}")
                if t is not None and (tuple(t.shape) != want or t.dtype != dt or not t.is_contiguous()):
                    raise RuntimeError(f"sample_keys: the {name} buffer {tuple(t.shape)} {t.dtype} (strides {t.stride()}) "
                                       f"does not fit {want} contiguous {dt}")
        nv = (ctypes.c_int32 * n_scenes)(*[int(v) for v in n_valid])
        wsb = lib.usc_sample_keys_ws_bytes(n_scenes, K, q) if q else 0
        ws = _ws(wsb, dev) if q else None
        check(lib.usc_sample_keys(_ptr(feats), c, _ptr(mask), q, _ptr(pos), p, _ptr(idx), n_scenes, K, nv, _ptr(of),
                                  _ptr(om), _ptr(op), _ptr(ws), 0 if ws is None else ws.numel(), _stream()), "usc_sample_keys")
        ctx.save_for_backward(idx)
        # the mask and positional outputs never carry a gradient: without this the engine hands backward() zero-filled
        # [B, K, Q] / [B, K, p] tensors for them (two stock fills per decoder pass, up to 26 us for the bool one)
        ctx.set_materialize_grads(False)
        ctx.n_src, ctx.unique = (0 if feats is None else feats.shape[0]), bool(unique)
        ctx.valid_unique = bool(valid_unique)
        ctx.K, ctx.n_valid = int(K), [min(int(v), int(K)) for v in n_valid]
        ctx.sink = sink if (sink is not None and ctx.needs_input_grad[0]) else None     # (forward runs under no_grad)
        if ctx.sink is not None:
            ctx.sink.pending += 1
        outs_ = tuple(t for t in (of, om, op) if t is not None)
        for t in (om, op):
            if t is not None:
                ctx.mark_non_differentiable(t)
        return outs_ if len(outs_) > 1 else outs_[0]

    @staticmethod
    def backward(ctx, dfeats, *_):
        (idx,) = ctx.saved_tensors
        sink = ctx.sink
        if dfeats is None:                     # nothing downstream used the sampled features
            if sink is not None:
                sink.pending -= 1
                if sink.pending == 0 and sink.buf is not None:
                    dsrc, sink.buf = sink.buf, None
                    return (dsrc,) + (None,) * 10
            return (None,) * 11
        dfeats = dfeats.contiguous().view(idx.shape[0], -1)
        c = dfeats.shape[1]

        def scatter(dst, add):
            # unique: all K keys of every scene are distinct rows.  valid_unique (the decoder's plans, models/mask3d.py
            # _draw_key_samples): the first n_valid[b] keys of a scene are distinct — a random subset, or every row of the
            # scene — and the rest is padding that repeats row 0 and is masked for every query, so its gradient rows are
            # exact zeros: only the distinct prefix is scattered, with plain (or read-modify-write) row stores.
            # (Scattering the padding with float atomics serialised on row 0: 214 us per pass on a 20 k-voxel scene.)
            # Neither: duplicates anywhere — float atomics over all keys.
            if not (ctx.unique or ctx.valid_unique):
                check(lib.usc_scatter_add_rows(_ptr(dfeats), c, _ptr(idx), idx.shape[0], _ptr(dst), _stream()),
                      "usc_scatter_add_rows")
                return
            fn = lib.usc_scatter_rows_unique_add if add else lib.usc_scatter_rows_unique
            if ctx.unique or all(v == ctx.K for v in ctx.n_valid):
                check(fn(_ptr(dfeats), c, _ptr(idx), idx.shape[0], _ptr(dst), _stream()), "usc_scatter_rows")
                return
            for b, nv in enumerate(ctx.n_valid):
                if nv > 0:
                    check(fn(dfeats.data_ptr() + 4 * b * ctx.K * c, c, idx.data_ptr() + 8 * b * ctx.K, nv, _ptr(dst),
                             _stream()), "usc_scatter_rows")

        if sink is not None:
            first = sink.buf is None
            if first:
                sink.buf = torch.zeros((ctx.n_src, c), dtype=torch.float32, device=dfeats.device)
                if not sink.check_queued:
                    sink.check_queued = True
                    torch.autograd.Variable._execution_engine.queue_callback(sink._check)
            scatter(sink.buf, add=not first)
            sink.pending -= 1
            if sink.pending > 0:
                return (None,) * 11
            dsrc, sink.buf = sink.buf, None
            return (dsrc,) + (None,) * 10
        dsrc = torch.zeros((ctx.n_src, c), dtype=torch.float32, device=dfeats.device)
        scatter(dsrc, add=False)
        return (dsrc,) + (None,) * 10


def sample_keys(feats, mask, pos, idx, n_scenes, K, n_valid, outs=None, unique=False, sink=None, valid_unique=False):
    """The cross-attention keys of one decoder pass (reference models/mask3d.py:306-346) in two launches:
    rows `idx` (i64[n_scenes*K], batch-wide row numbers) of the level's features f32[n,c], thresholded attention
    masks bool[n,Q] and positional encodings f32[n,p] (or None) -> ([B,K,c], bool[B,K,Q], [B,K,p]) — feats None: the
    masks only (-> bool[B,K,Q]); mask None: the features (and positions) only, in one launch; a query column
    masked in all K rows of its scene is cleared; rows k >= n_valid[b] (padding) are fully masked.
    outs: the caller's three buffers (e.g. the inputs of a captured pass).  Gradient: features only (scatter;
    `unique` as in gather_rows; valid_unique: only the first n_valid[b] keys of every scene are distinct and the padding
    behind them carries no gradient).  sink: a GradSink shared by every call that samples the SAME `feats` in this forward
    pass — their gradients are then accumulated into one buffer."""
    if feats is None and mask is None:
        raise RuntimeError("sample_keys: nothing to gather")
    rows = None
    for t, dt, name in ((feats, torch.float32, "feats"), (mask, torch.bool, "mask"), (pos, torch.float32, "pos")):
        if t is not None:
            _chk(t, dt, name)
            if rows is not None and t.shape[0] != rows:
                raise RuntimeError("sample_keys: feats, mask and pos must have the same rows")
            rows = t.shape[0]
    _chk(idx, torch.int64, "idx")
    if idx.shape[0] != n_scenes * K or len(n_valid) != n_scenes:
        raise RuntimeError("sample_keys: inconsistent sizes")
    return _SampleKeys.apply(feats, mask, pos, idx, int(n_scenes), int(K), list(n_valid), outs, unique, sink,
                             valid_unique)


def gather_rows_i32(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """Row gather of an int32 table (coordinates) through the same kernel (bit pattern copy)."""
    _chk(src, torch.int32, "src")
    return _GatherRows.apply(src.view(torch.float32), idx.contiguous(), None, False).view(torch.int32)


@dataclass
class SegmentCSR:
    seg: torch.Tensor      # i64[n]
    order: torch.Tensor    # i64[n] rows sorted by segment (stable)
    seg_off: torch.Tensor  # i64[S+1]
    S: int


def segment_csr(seg: torch.Tensor, S: int) -> SegmentCSR:
    _chk(seg, torch.int64, "seg")
    n = seg.shape[0]
    dev = seg.device
    order = torch.empty(n, dtype=torch.int64, device=dev)
    seg_off = torch.empty(S + 1, dtype=torch.int64, device=dev)
    ws = _ws(lib.usc_segment_csr_ws_bytes(n, S), dev)
    check(lib.usc_segment_csr(_ptr(seg), n, S, _ptr(order), _ptr(seg_off), _ptr(ws), ws.numel(), _stream()),
          "usc_segment_csr")
    return SegmentCSR(seg, order, seg_off, S)


def segment_sum_from_csr(src, csr: SegmentCSR):
    mean = torch.empty((csr.S, src.shape[1]), dtype=torch.float32, device=src.device)
    check(lib.usc_segment_mean_fwd(_ptr(src), src.shape[1], _ptr(csr.order), _ptr(csr.seg_off), csr.S, _ptr(mean),
                                   _stream()), "usc_segment_mean_fwd")
    cnt = (csr.seg_off[1:] - csr.seg_off[:-1]).to(torch.float32)
    return mean * cnt[:, None]


class _SegmentMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, csr):
        src = src.contiguous()
        out = torch.empty((csr.S, src.shape[1]), dtype=torch.float32, device=src.device)
        check(lib.usc_segment_mean_fwd(_ptr(src), src.shape[1], _ptr(csr.order), _ptr(csr.seg_off), csr.S, _ptr(out),
                                       _stream()), "usc_segment_mean_fwd")
        ctx.csr, ctx.n = csr, src.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        csr = ctx.csr
        dsrc = torch.empty((ctx.n, dout.shape[1]), dtype=torch.float32, device=dout.device)
        check(lib.usc_segment_mean_bwd(_ptr(dout), dout.shape[1], _ptr(csr.seg), _ptr(csr.seg_off), ctx.n,
                                       _ptr(dsrc), _stream()), "usc_segment_mean_bwd")
        return dsrc, None


def segment_mean(src, csr: SegmentCSR):
    """torch_scatter.scatter_mean(src, seg, dim=0) with a prebuilt CSR (reference models/mask3d.py:223)."""
    _chk(src, torch.float32, "src")
    return _SegmentMean.apply(src, csr)


# ------------------------------------------------------------------ points
def furthest_point_sample(xyz: torch.Tensor, npoint: int) -> torch.Tensor:
    """pointnet2_utils.furthest_point_sample(xyz f32[B,N,3], npoint) -> i32[B,npoint]
    (reference third_party/pointnet2/pointnet2_utils.py:50-79)."""
    require_device()
    _chk(xyz, torch.float32, "xyz")
    if xyz.dim() != 3 or xyz.shape[2] != 3:
        raise RuntimeError("xyz must be f32 [B,N,3]")
    B, N, _ = xyz.shape
    tmp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
    idx = torch.zeros((B, npoint), dtype=torch.int32, device=xyz.device)
    check(lib.usc_furthest_point_sampling(_ptr(xyz), B, N, npoint, _ptr(tmp), _ptr(idx), _stream()),
          "usc_furthest_point_sampling")
    return idx


def fourier_posenc(xyz, lo, hi, gauss_B, d):
    """PositionEmbeddingCoordsSine('fourier', normalize=True): f32[n,3] -> f32[n,d]."""
    _chk(xyz, torch.float32, "xyz")
    out = torch.empty((xyz.shape[0], d), dtype=torch.float32, device=xyz.device)
    check(lib.usc_fourier_posenc(_ptr(xyz), xyz.shape[0], _ptr(lo.contiguous()), _ptr(hi.contiguous()),
                                 _ptr(gauss_B.contiguous()), d, _ptr(out), _stream()), "usc_fourier_posenc")
    return out


# ------------------------------------------------------------------ neighbours / clustering
def knn1(query: torch.Tensor, ref: torch.Tensor):
    """Exact nearest reference point of every query (scipy KDTree.query(k=1)): -> (dist f32[nq], idx i64[nq])."""
    require_device()
    _chk(query, torch.float32, "query")
    _chk(ref, torch.float32, "ref")
    nq = query.shape[0]
    idx = torch.empty(nq, dtype=torch.int64, device=query.device)
    d2 = torch.empty(nq, dtype=torch.float32, device=query.device)
    check(lib.usc_knn1(_ptr(query), nq, _ptr(ref), ref.shape[0], _ptr(idx), _ptr(d2), _stream()), "usc_knn1")
    return d2.sqrt(), idx


def cc_eps(xyz: torch.Tensor, eps: float, max_rounds: int = 256) -> torch.Tensor:
    """Connected components of the eps-ball graph == DBSCAN(eps, min_samples=1).labels_ (first-seen order)."""
    require_device()
    _chk(xyz, torch.float32, "xyz")
    n = xyz.shape[0]
    dev = xyz.device
    la = torch.empty(n, dtype=torch.int32, device=dev)
    lb = torch.empty(n, dtype=torch.int32, device=dev)
    changed = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib.usc_cc_eps_init(_ptr(la), n, _stream()), "usc_cc_eps_init")
    for _ in range(max_rounds):
        check(lib.usc_cc_eps_step(_ptr(xyz), n, float(eps), _ptr(la), _ptr(lb), _ptr(changed), _stream()),
              "usc_cc_eps_step")
        la, lb = lb, la
        if int(changed.item()) == 0:
            break
    else:
        raise RuntimeError("cc_eps: label propagation did not converge")
    labels = torch.empty(n, dtype=torch.int64, device=dev)
    check(lib.usc_cc_eps_finish(_ptr(la), n, _ptr(lb), _ptr(labels), _stream()), "usc_cc_eps_finish")
    return labels


# ------------------------------------------------------------------ decoder-side small-row ops
_LN_DIMS = {64, 128, 192, 256, 384, 512}


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, passthrough=False):
        d = x.shape[-1]
        x2 = x.contiguous().view(-1, d)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        check(lib.usc_layernorm_fwd(_ptr(x2), _ptr(weight), _ptr(bias), rows, d, float(eps), _ptr(y), _ptr(mean),
                                    _ptr(rstd), _stream()), "usc_layernorm_fwd")
        ctx.save_for_backward(x2, weight, mean, rstd)
        ctx.shape = x.shape
        ctx.w_param, ctx.b_param = weight, bias
        ctx.passthrough = bool(passthrough)
        # passthrough: x comes back as a second output; the gradient that arrives there (x's other consumer) is summed
        # inside the backward launch (usc_layernorm_bwd_ex: dx_add) instead of by a separate autograd add
        return (y.view(x.shape), x) if passthrough else y.view(x.shape)

    @staticmethod
    def backward(ctx, dy, dpass=None):
        x2, weight, mean, rstd = ctx.saved_tensors
        rows, d = x2.shape
        if dy is None:                                  # only the passthrough output was used
            return (None if dpass is None else dpass), None, None, None, None
        dy2 = dy.contiguous().view(rows, d)
        dadd = None if dpass is None else dpass.contiguous().view(rows, d)
        # x is the output buffer of a captured decoder pass: its gradient goes straight into the buffer the pass's
        # backward graph reads (graphs.grad_buffer_for), which saves the copy in front of the replay.  Safe whether or
        # not this norm is x's only consumer: if autograd has to add another consumer's gradient, the sum is a new
        # tensor and _GraphedFn.backward copies it over what was written here before it replays.
        from .graphs import grad_buffer_for
        buf = grad_buffer_for(x2)
        dx = buf.view(rows, d) if buf is not None else torch.empty_like(x2)
        tg, tb = _grad_target(ctx.w_param), _grad_target(ctx.b_param)
        in_place = tg is not None and tb is not None
        dgamma = tg if in_place else torch.empty_like(weight)
        dbeta = tb if in_place else torch.empty_like(weight)
        wsb = lib.usc_layernorm_bwd_ws_bytes(rows, d)
        ws = _ws(wsb, x2.device) if wsb > 0 else None
        check(lib.usc_layernorm_bwd_ex(_ptr(dy2), _ptr(x2), _ptr(mean), _ptr(rstd), _ptr(weight), rows, d, _ptr(dadd),
                                       _ptr(dx), _ptr(dgamma), _ptr(dbeta), int(in_place), _ptr(ws), wsb, _stream()),
              "usc_layernorm_bwd")
        if in_place:
            dgamma = dbeta = None
            _grad_written(ctx.w_param, ctx.b_param)
        return dx.view(ctx.shape), dgamma, dbeta, None, None


class _AddLayerNorm(torch.autograd.Function):
    """LayerNorm(x + res) in one launch; both addends receive the same input gradient (no extra kernel)."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, eps):
        d = x.shape[-1]
        x2 = x.contiguous().view(-1, d)
        r2 = res.contiguous().view(-1, d)
        rows = x2.shape[0]
        y, ssum = torch.empty_like(x2), torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        check(lib.usc_add_layernorm_fwd(_ptr(x2), _ptr(r2), _ptr(weight), _ptr(bias), rows, d, float(eps), _ptr(y),
                                        _ptr(ssum), _ptr(mean), _ptr(rstd), _stream()), "usc_add_layernorm_fwd")
        ctx.save_for_backward(ssum, weight, mean, rstd)
        ctx.shape = x.shape
        ctx.w_param, ctx.b_param = weight, bias
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        dx, dgamma, dbeta, _, _ = _LayerNorm.backward(ctx, dy)
        return dx, dx, dgamma, dbeta, None


def add_layer_norm(x, res, weight, bias, eps=1e-5):
    """F.layer_norm(x + res) (the post-norm residual of the decoder layers, reference models/mask3d.py:523-524)."""
    _chk(weight, torch.float32, "weight")
    _chk(bias, torch.float32, "bias")
    if x.dtype != torch.float32 or not x.is_cuda or x.shape[-1] not in _LN_DIMS or x.shape != res.shape:
        raise RuntimeError(f"add_layer_norm: needs two equal-shape f32 HIP tensors with last dim in {sorted(_LN_DIMS)}")
    return _AddLayerNorm.apply(x, res, weight, bias, eps)


def layer_norm(x, weight, bias, eps=1e-5, passthrough=False):
    """F.layer_norm over the last dimension through the HIP kernels (f32, contiguous weight/bias, d in _LN_DIMS).
    passthrough: -> (y, x'), x' = x as an output of the same autograd node: hand it to x's OTHER consumer and that
    consumer's gradient is summed inside this norm's backward launch."""
    _chk(weight, torch.float32, "weight")
    _chk(bias, torch.float32, "bias")
    if x.dtype != torch.float32 or not x.is_cuda or x.shape[-1] not in _LN_DIMS:
        raise RuntimeError(f"layer_norm: needs an f32 HIP tensor with last dim in {sorted(_LN_DIMS)}")
    return _LayerNorm.apply(x, weight, bias, eps, passthrough)


# linear layers with at most this many rows go to the tile-per-workgroup kernels of decoder.hip (the 100 queries; the
# few hundred to a thousand segments of the mask logits: 5 us per launch against 11-21 us on the many-row kernels)
SMALL_ROWS = 1024


def _small_linear_ok(rows, n_in, n_out):
    return rows <= SMALL_ROWS and n_in % 32 == 0 and n_out % 32 == 0 and n_in >= 32 and n_out >= 32


def _rows_gemm_ok(rows, n_in, n_out):
    """Many-row linear layers (the sampled voxels of a decoder pass: 200 ... 12 800 rows; segment logits) run on the
    1x1-convolution kernels — the library's heuristics pick 30-95 us kernels for these 0.1-0.4 GFLOP shapes."""
    return rows > SMALL_ROWS and n_in % 32 == 0 and n_out % 32 == 0


def _lin_fwd(x2, W, b, add=None, relu=False, pad_rows_to=None, out=None):
    """y = (x2 [+ add]) W^T + b [then ReLU] for contiguous f32 x2 [M,K], W [N,K] (a contiguous row block is fine);
    `add` / `relu` are folded into the launch on the few-row kernels and applied separately elsewhere.
    pad_rows_to: y gets that many rows, the extra ones zero (few-row kernels only)."""
    M, K = x2.shape
    N = W.shape[0]
    if _small_linear_ok(M, K, N):
        Mp = M if pad_rows_to is None else int(pad_rows_to)
        y = torch.empty((Mp, N), dtype=torch.float32, device=x2.device) if out is None else out
        if tuple(y.shape) != (Mp, N) or y.dtype != torch.float32 or not y.is_contiguous():
            raise RuntimeError(f"linear: `out` must be a contiguous f32 [{Mp}, {N}] tensor")
        check(lib.usc_linear_fwd_pad(_ptr(x2), _ptr(add), _ptr(W), _ptr(b), M, N, K, int(relu), _ptr(y), Mp, _stream()),
              "usc_linear_fwd")
        return y
    if pad_rows_to is not None:
        raise RuntimeError("linear: pad_rows_to needs the few-row kernels (rows <= 1024, widths multiples of 32)")
    if add is not None:
        x2 = x2 + add
    if relu:
        return torch.relu_(_lin_fwd(x2, W, b, out=out))
    if _rows_gemm_ok(M, K, N) and W.is_contiguous():
        # W [N, K] is the product's [cout][cin]: read in place by the row-order kernel (was: a transpose launch per call,
        # 36 per training step inside the captured decoder passes)
        return gather_gemm(x2, W.view(1, N, K), None, M, bias=b, out=out, w_transposed=True)
    y = torch.addmm(b, x2, W.t()) if b is not None else x2 @ W.t()
    return y if out is None else out.copy_(y)


def col_sum(x2, out, accumulate=False):
    """out[c] (+)= sum_r x2[r, c] in a fixed order (usc_col_sum).  Not torch.sum: replayed from the captured decoder
    graphs, ATen's multi-block reduction left the bias gradients of the 3 200 / 12 800-row projections at the value
    of the previous reduction that had used the same scratch block (found by the padded-level graph test)."""
    _chk(x2, torch.float32, "x2")
    if not out.is_contiguous() or out.dtype != torch.float32 or out.numel() != x2.shape[1]:
        raise RuntimeError("col_sum: `out` must be a contiguous f32 vector with one entry per column")
    wsb = lib.usc_col_sum_ws_bytes(x2.shape[0], x2.shape[1])
    ws = _ws(wsb, x2.device)
    check(lib.usc_col_sum(_ptr(x2), x2.shape[0], x2.shape[1], _ptr(out), int(accumulate), _ptr(ws), wsb, _stream()),
          "usc_col_sum")
    return out


def _lin_bwd(dy2, x2, W, dW_out, db_out, need_dx=True, accumulate=False, add=None, y_relu=None, dx_add=None,
             dx_add2=None, want_dx_b=False):
    """-> dx (or None); writes (accumulate: adds) dW_out [N,K] and db_out [N] (row-block views are fine).
    add: the layer's input was x2 + add; y_relu: its output after the fused ReLU; dx_add: added to dx.
    dx_add2 / want_dx_b: -> (dx, dx_b) with dx = dy W + dx_add + dx_add2 and dx_b = dy W + dx_add (usc_linear_bwd_ex2)."""
    M, N = dy2.shape
    K = x2.shape[1]
    if want_dx_b or dx_add2 is not None:
        if not need_dx:
            return (None, None) if want_dx_b else None
        if _small_linear_ok(M, K, N):
            dx = torch.empty((M, K), dtype=torch.float32, device=dy2.device)
            dx_b = torch.empty_like(dx) if want_dx_b else None
            check(lib.usc_linear_bwd_ex2(_ptr(dy2), _ptr(y_relu), _ptr(x2), _ptr(add), _ptr(W), M, N, K, _ptr(dx),
                                         _ptr(dx_add), _ptr(dx_add2), _ptr(dx_b), _ptr(dW_out), _ptr(db_out),
                                         int(accumulate), _stream()), "usc_linear_bwd")
        else:
            dx_b = _lin_bwd(dy2, x2, W, dW_out, db_out, True, accumulate, add, y_relu, dx_add)
            dx = dx_b if dx_add2 is None else dx_b + dx_add2
        return (dx, dx_b) if want_dx_b else dx
    if _small_linear_ok(M, K, N):
        dx = torch.empty((M, K), dtype=torch.float32, device=dy2.device) if need_dx else None
        check(lib.usc_linear_bwd_ex(_ptr(dy2), _ptr(y_relu), _ptr(x2), _ptr(add), _ptr(W), M, N, K, _ptr(dx),
                                    _ptr(dx_add) if need_dx else None, _ptr(dW_out), _ptr(db_out), int(accumulate),
                                    _stream()), "usc_linear_bwd")
        return dx
    if y_relu is not None:
        dy2 = dy2 * (y_relu > 0)
    if add is not None:
        x2 = x2 + add
    if dx_add is not None:
        dx = _lin_bwd(dy2, x2, W, dW_out, db_out, need_dx, accumulate)
        return None if dx is None else dx.add_(dx_add)
    rows_ok = _rows_gemm_ok(M, K, N) and W.is_contiguous()
    if rows_ok and dW_out.is_contiguous():
        # dW[N,K] = dy^T x over many rows: the weight-gradient kernel with identity pairs (a = dy, b = x)
        if accumulate:
            wgrad(dy2, x2, 1, into=dW_out)
        else:
            wgrad(dy2, x2, 1, out=dW_out)
    elif accumulate:
        dW_out.addmm_(dy2.t(), x2)
    else:
        torch.mm(dy2.t(), x2, out=dW_out)
    if db_out is not None:
        col_sum(dy2, db_out, accumulate)
    if not need_dx:
        return None
    if rows_ok:
        return gather_gemm(dy2, W.view(1, N, K), None, M)                # W [N, K] is this product's [cin, cout]
    return dy2 @ W


class _LinearRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, relu=False, passthrough=False, pad_rows_to=None):
        x2 = x.contiguous().view(-1, x.shape[-1])
        ctx.rows = x2.shape[0]
        if pad_rows_to is not None:
            if relu or passthrough or x2.shape[0] != x.shape[-2] or pad_rows_to < x2.shape[0]:
                raise RuntimeError("linear: pad_rows_to is for one [.., M, K] table, M <= pad_rows_to, without relu")
            y = _lin_fwd(x2, W, b, pad_rows_to=pad_rows_to)
            ctx.save_for_backward(x2, W, None)
            ctx.has_bias, ctx.shape = b is not None, x.shape
            ctx.w_param, ctx.b_param = W, b
            return y.view(*x.shape[:-2], int(pad_rows_to), W.shape[0])
        y = _lin_fwd(x2, W, b, relu=relu)
        ctx.save_for_backward(x2, W, y if relu else None)
        ctx.has_bias, ctx.shape = b is not None, x.shape
        ctx.w_param, ctx.b_param = W, b
        y = y.view(*x.shape[:-1], W.shape[0])
        # passthrough: x is handed back as a second output; whatever gradient arrives there (the residual path around
        # the layer) is added inside the input-gradient launch instead of by a separate autograd add
        return (y, x) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x2, W, y_relu = ctx.saved_tensors
        dy2 = dy.contiguous().view(-1, W.shape[0])[:ctx.rows]       # [:rows]: drops the zero-extension rows, if any
        tw = _grad_target(ctx.w_param)
        tb = _grad_target(ctx.b_param) if ctx.has_bias else None
        in_place = tw is not None and (tb is not None or not ctx.has_bias)
        if in_place:
            dW, db = tw, tb
        else:
            dW = torch.empty_like(W)
            db = torch.empty(W.shape[0], dtype=torch.float32, device=W.device) if ctx.has_bias else None
        dres2 = None if dres is None else dres.contiguous().view(-1, x2.shape[1])
        dx = _lin_bwd(dy2, x2, W, dW, db, need_dx=ctx.needs_input_grad[0], accumulate=in_place, y_relu=y_relu,
                      dx_add=dres2)
        if in_place:
            dW = db = None
            _grad_written(ctx.w_param, ctx.b_param if ctx.has_bias else None)
        return (None if dx is None else dx.view(ctx.shape)), dW, db, None, None, None


def linear(x, W, b=None, relu=False, passthrough=False, pad_rows_to=None):
    """F.linear(x, W, b) (relu: followed by ReLU, in the same launch on the few-row kernels) for f32 HIP tensors;
    few-row inputs run on the wave-per-tile MFMA kernels of decoder.hip.
    passthrough: -> (y, x'), x' = x as an output of the same autograd node: use it for the residual connection
    around the layer, its gradient is then summed inside the layer's input-gradient launch.
    pad_rows_to: x is ONE table [.., M, K]; y comes back as [.., pad_rows_to, N] with zero rows appended (the
    gradient of those rows is dropped)."""
    _chk(W, torch.float32, "W")
    return _LinearRows.apply(x, W, b, relu, passthrough, pad_rows_to)


FUSED_QKV = os.environ.get("USC3D_FUSED_QKV", "1") == "1"


class _InProj(torch.autograd.Function):
    """q, k, v = the three input projections of nn.MultiheadAttention from its packed in_proj_weight [3E,E] /
    in_proj_bias [3E] (reference: nn.MultiheadAttention inside models/mask3d.py:491-605).  One Function so that the
    three weight gradients are written into ONE [3E,E] tensor instead of three sliced ones summed by autograd.
    pos_q / pos_k (optional): q = (xq + pos_q) Wq, k = (xk + pos_k) Wk — the `with_pos_embed` adds of the reference
    (:485, :517) folded into the projection launches.  Inputs that are the SAME tensor (self attention: xq, xk, xv;
    cross attention: xk, xv) get one summed gradient, chained through the input-gradient launches (dx_add) instead of
    separate tensors added by autograd."""

    @staticmethod
    def forward(ctx, xq, xk, xv, W, b, pos_q, pos_k, residual=False):
        E = W.shape[1]
        xs = [t.contiguous().view(-1, E) for t in (xq, xk, xv)]
        ps = [None if t is None else t.contiguous().view(-1, E) for t in (pos_q, pos_k, None)]
        for j in range(2):     # many-row inputs (the sampled voxels): no fused add on those kernels, keep the sum
            if ps[j] is not None and not _small_linear_ok(xs[j].shape[0], E, E):
                xs[j], ps[j] = xs[j] + ps[j], None
        def same_t(a, c):
            return a.data_ptr() == c.data_ptr() and a.shape == c.shape
        if (FUSED_QKV and same_t(xs[0], xs[1]) and same_t(xs[1], xs[2]) and _small_linear_ok(xs[0].shape[0], E, 3 * E)
                and W.is_contiguous() and (ps[0] is None) == (ps[1] is None)
                and (ps[0] is None or same_t(ps[0], ps[1]))):
            # self attention: one launch for the three projections of the one input ([3, M, E], each its own matrix)
            M = xs[0].shape[0]
            y3 = torch.empty((3, M, E), dtype=torch.float32, device=W.device)
            check(lib.usc_linear_fwd_split(_ptr(xs[0]), _ptr(ps[0]), _ptr(W), _ptr(b), M, 3 * E, E, 2 * E, E, _ptr(y3),
                                           _stream()), "usc_linear_fwd_split")
            outs = [y3[0], y3[1], y3[2]]
        else:
            outs = [_lin_fwd(xs[j], W[j * E:(j + 1) * E], b[j * E:(j + 1) * E], add=ps[j]) for j in range(3)]
        ctx.save_for_backward(xs[0], xs[1], xs[2], W, ps[0], ps[1])
        ctx.shapes = (xq.shape, xk.shape, xv.shape)
        ctx.pos_shapes = (None if pos_q is None else pos_q.shape, None if pos_k is None else pos_k.shape)
        ctx.w_param, ctx.b_param = W, b

        def same(a, c):
            return a is not None and c is not None and a.data_ptr() == c.data_ptr() and a.shape == c.shape \
                and a.stride() == c.stride()
        # q and k may share one summed input gradient only if their positional terms are the same too (both absent,
        # or one tensor) — also when a term was folded into the many-row input above
        ctx.same_qk, ctx.same_kv = same(xq, xk), same(xk, xv)
        ctx.same_pos = (pos_q is None and pos_k is None) or same(pos_q, pos_k)
        out = tuple(o.view(*shp[:-1], E) for o, shp in zip(outs, ctx.shapes))
        # residual: xq is handed back as a fourth output (the `tgt + attention(tgt ...)` residual of the decoder
        # blocks); its gradient joins the query input's inside the input-gradient launches
        return out + (xq,) if residual else out

    @staticmethod
    def backward(ctx, dq, dk, dv, dres=None):
        x0, x1, x2, W, p0, p1 = ctx.saved_tensors
        E = W.shape[1]
        dres2 = None if dres is None else dres.contiguous().view(-1, E)
        tw, tb = _grad_target(ctx.w_param), _grad_target(ctx.b_param)
        in_place = tw is not None and tb is not None
        if in_place:
            dW, db = tw, tb
        else:
            dW = torch.empty_like(W)
            db = torch.empty(3 * E, dtype=torch.float32, device=W.device)
        need = list(ctx.needs_input_grad[:3])
        need_pq, need_pk = ctx.needs_input_grad[5], ctx.needs_input_grad[6]

        def bwd(j, dyj, xj, pj, need_dx, dx_add=None):
            return _lin_bwd(dyj.contiguous().view(-1, E), xj, W[j * E:(j + 1) * E], dW[j * E:(j + 1) * E],
                            db[j * E:(j + 1) * E], need_dx=need_dx, accumulate=in_place, add=pj, dx_add=dx_add)

        M = x0.shape[0]
        adjacent = (dq.is_contiguous() and dk.is_contiguous() and dv.is_contiguous()
                    and dk.data_ptr() == dq.data_ptr() + 4 * M * E and dv.data_ptr() == dk.data_ptr() + 4 * M * E)
        if (FUSED_QKV and ctx.same_qk and ctx.same_kv and ctx.same_pos and need[0] and adjacent and W.is_contiguous()
                and _small_linear_ok(M, 3 * E, E) and dW.is_contiguous() and db.is_contiguous()):
            # self attention: the three projections backwards in one launch (usc_qkv_proj_bwd)
            gx = torch.empty((M, E), dtype=torch.float32, device=W.device)
            gpos = torch.empty_like(gx) if (need_pq and p0 is not None) else None
            check(lib.usc_qkv_proj_bwd(_ptr(dq), _ptr(x0), _ptr(p0), _ptr(W), M, E, _ptr(gx), _ptr(gpos), _ptr(dres2),
                                       _ptr(dW), _ptr(db), int(in_place), _stream()), "usc_qkv_proj_bwd")
            out_x = (gx, None, None)
            out_p = (gpos, None)
        elif dres2 is not None and ctx.same_qk and ctx.same_kv and ctx.same_pos and need[0]:
            # self attention with the residual: v (+ residual) first, then k, then q on top of both; the q launch also
            # writes the sum WITHOUT the v/residual part — the positional term's gradient
            gv = bwd(2, dv, x2, None, True, dx_add=dres2)
            gk = bwd(1, dk, x1, p1, True)
            gx, gpos = _lin_bwd(dq.contiguous().view(-1, E), x0, W[:E], dW[:E], db[:E], need_dx=True,
                                accumulate=in_place, add=p0, dx_add=gk, dx_add2=gv, want_dx_b=True)
            out_x = (gx, None, None)
            out_p = (gpos if need_pq else None, None)
        elif dres2 is not None and not ctx.same_qk and need[0]:
            # cross attention with the residual: the query product feeds d(tgt) (+ residual) and d(query_pos)
            gk = bwd(1, dk, x1, p1, need[1] or need_pk)
            gq, gq_pos = _lin_bwd(dq.contiguous().view(-1, E), x0, W[:E], dW[:E], db[:E], need_dx=True,
                                  accumulate=in_place, add=p0, dx_add2=dres2, want_dx_b=True)
            chain_v = ctx.same_kv and gk is not None
            gv = bwd(2, dv, x2, None, need[2], dx_add=gk if chain_v else None)
            out_x = (gq, gv, None) if chain_v else (gq, gk, gv)
            out_p = (gq_pos if need_pq else None, gk if need_pk else None)
        else:
            out_x, out_p = _InProj._backward_plain(ctx, bwd, dq, dk, dv, x0, x1, x2, p0, p1, need, need_pq, need_pk)
            if dres2 is not None and need[0]:          # shapes the fused forms above do not cover
                out_x = (dres2 if out_x[0] is None else out_x[0] + dres2,) + tuple(out_x[1:])
        if in_place:
            dW = db = None
            _grad_written(ctx.w_param, ctx.b_param)
        gx = tuple(None if (g is None or not n) else g.view(shp) for g, n, shp in zip(out_x, need, ctx.shapes))
        gp = tuple(None if g is None else g.view(shp) for g, shp in zip(out_p, ctx.pos_shapes))
        return gx + (dW, db) + gp + (None,)

    @staticmethod
    def _backward_plain(ctx, bwd, dq, dk, dv, x0, x1, x2, p0, p1, need, need_pq, need_pk):
        # k first, then q on top of it when they share the input (and the positional term); v last on top of both
        # when it shares the input too: one summed tensor per distinct input
        gk = bwd(1, dk, x1, p1, need[1] or need_pk)
        chain_q = ctx.same_qk and gk is not None and ctx.same_pos
        gq = bwd(0, dq, x0, p0, need[0] or need_pq, dx_add=gk if chain_q else None)
        if chain_q:                       # gq = d(xq + pos) summed over the q and k paths
            gx, gpos = gq, gq
            gv = bwd(2, dv, x2, None, need[2], dx_add=gx if ctx.same_kv else None)
            if ctx.same_kv:
                out_x = (gv, None, None)
            else:
                out_x = (gx, None, gv)
            out_p = (gpos if need_pq else None, None)
        else:
            chain_v = ctx.same_kv and gk is not None
            gv = bwd(2, dv, x2, None, need[2], dx_add=gk if chain_v else None)
            out_x = (gq, gv, None) if chain_v else (gq, gk, gv)
            out_p = (gq if need_pq else None, gk if need_pk else None)
        return out_x, out_p


def _packed_grad_targets(W, b):
    """(dW [3E,E], db [3E], in_place) for one part of a packed in-projection: the parameters' gradient buffers when they
    exist (the part's rows are added into), else fresh zero tensors (returned to autograd whole)."""
    tw, tb = _grad_target(W), _grad_target(b)
    if tw is not None and tb is not None:
        return tw, tb, True
    return torch.zeros_like(W), torch.zeros_like(b), False


class _InProjQ(torch.autograd.Function):
    """q = (x + pos) Wq^T + bq, Wq = in_proj_weight[:E] — the QUERY third of nn.MultiheadAttention's packed input
    projection on its own (reference models/mask3d.py:547-605 CrossAttentionLayer, :517 with_pos_embed), so that the key /
    value thirds can be computed elsewhere (ops.in_proj_kv: they do not depend on the queries).  residual: x comes back
    as a second output and its gradient (the block's residual path) is summed inside the input-gradient launch."""

    @staticmethod
    def forward(ctx, x, W, b, pos, residual=False):
        E = W.shape[1]
        x2 = x.contiguous().view(-1, E)
        p2 = None if pos is None else pos.contiguous().view(-1, E)
        if p2 is not None and not _small_linear_ok(x2.shape[0], E, E):
            x2, p2 = x2 + p2, None
        y = _lin_fwd(x2, W[:E], b[:E], add=p2)
        ctx.save_for_backward(x2, W, p2)
        ctx.shape, ctx.pos_shape = x.shape, None if pos is None else pos.shape
        ctx.w_param, ctx.b_param = W, b
        y = y.view(*x.shape[:-1], E)
        return (y, x) if residual else y

    @staticmethod
    def backward(ctx, dq, dres=None):
        x2, W, p2 = ctx.saved_tensors
        E = W.shape[1]
        dW, db, in_place = _packed_grad_targets(ctx.w_param, ctx.b_param)
        dq2 = dq.contiguous().view(-1, E)
        dres2 = None if dres is None else dres.contiguous().view(-1, E)
        need_pos = ctx.pos_shape is not None and ctx.needs_input_grad[3]
        if dres2 is not None and need_pos:
            gx, gpos = _lin_bwd(dq2, x2, W[:E], dW[:E], db[:E], need_dx=True, accumulate=in_place, add=p2, dx_add2=dres2,
                                want_dx_b=True)
        else:
            gx = _lin_bwd(dq2, x2, W[:E], dW[:E], db[:E], need_dx=True, accumulate=in_place, add=p2, dx_add=dres2)
            gpos = gx if need_pos else None
        if in_place:
            dW = db = None
            _grad_written(ctx.w_param, ctx.b_param)
        return (gx.view(ctx.shape), dW, db, None if gpos is None or ctx.pos_shape is None else gpos.view(ctx.pos_shape), None)


class _InProjKV(torch.autograd.Function):
    """k = (x + pos) Wk^T + bk, v = x Wv^T + bv with Wk, Wv = in_proj_weight[E:2E], [2E:3E]: the key / value thirds of the
    packed input projection over the sampled voxels of a decoder pass.  `outs` (optional): (k, v) buffers to write into
    (the static inputs of a captured pass).  One summed gradient for x (k's part chained into v's launch)."""

    @staticmethod
    def forward(ctx, x, W, b, pos, outs=None):
        E = W.shape[1]
        x2 = x.contiguous().view(-1, E)
        p2 = None if pos is None else pos.contiguous().view(-1, E)
        xk, pk = x2, p2
        if pk is not None and not _small_linear_ok(x2.shape[0], E, E):
            xk, pk = x2 + pk, None                  # (many rows: no fused add on those kernels)
        ok, ov = (None, None) if outs is None else (outs[0].detach().view(-1, E), outs[1].detach().view(-1, E))
        k = _lin_fwd(xk, W[E:2 * E], b[E:2 * E], add=pk, out=ok)
        v = _lin_fwd(x2, W[2 * E:], b[2 * E:], out=ov)
        ctx.save_for_backward(xk, x2, W, pk)
        ctx.shape = x.shape
        ctx.w_param, ctx.b_param = W, b
        shp = (*x.shape[:-1], E)
        return k.view(shp), v.view(shp)

    @staticmethod
    def backward(ctx, dk, dv):
        xk, x2, W, pk = ctx.saved_tensors
        E = W.shape[1]
        dW, db, in_place = _packed_grad_targets(ctx.w_param, ctx.b_param)
        need = ctx.needs_input_grad[0]
        gk = _lin_bwd(dk.contiguous().view(-1, E), xk, W[E:2 * E], dW[E:2 * E], db[E:2 * E], need_dx=need,
                      accumulate=in_place, add=pk)
        gx = _lin_bwd(dv.contiguous().view(-1, E), x2, W[2 * E:], dW[2 * E:], db[2 * E:], need_dx=need,
                      accumulate=in_place, dx_add=gk)
        if in_place:
            dW = db = None
            _grad_written(ctx.w_param, ctx.b_param)
        return (None if gx is None else gx.view(ctx.shape)), dW, db, None, None


def in_proj_q(x, W, b, pos=None, residual=False):
    _chk(W, torch.float32, "in_proj_weight")
    _chk(b, torch.float32, "in_proj_bias")
    if pos is not None and pos.shape != x.shape:
        raise RuntimeError(f"in_proj_q: pos {tuple(pos.shape)} must have the shape of the input {tuple(x.shape)}")
    return _InProjQ.apply(x, W, b, pos, residual)


def in_proj_kv(x, W, b, pos=None, outs=None):
    _chk(W, torch.float32, "in_proj_weight")
    _chk(b, torch.float32, "in_proj_bias")
    if pos is not None and pos.shape != x.shape:
        raise RuntimeError(f"in_proj_kv: pos {tuple(pos.shape)} must have the shape of the input {tuple(x.shape)}")
    return _InProjKV.apply(x, W, b, pos, outs)


def in_proj(xq, xk, xv, W, b, pos_q=None, pos_k=None, residual=False):
    _chk(W, torch.float32, "in_proj_weight")
    _chk(b, torch.float32, "in_proj_bias")
    # the kernels read the positional term row for row: no broadcasting (a [Q,1,E] term with B > 1 would be read
    # out of bounds)
    if pos_q is not None and pos_q.shape != xq.shape:
        raise RuntimeError(f"in_proj: pos_q {tuple(pos_q.shape)} must have the shape of the query input {tuple(xq.shape)}")
    if pos_k is not None and pos_k.shape != xk.shape:
        raise RuntimeError(f"in_proj: pos_k {tuple(pos_k.shape)} must have the shape of the key input {tuple(xk.shape)}")
    return _InProj.apply(xq, xk, xv, W, b, pos_q, pos_k, residual)


class _MaskedCrossAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, num_heads):
        L, B, E = q.shape
        S = k.shape[0]
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        m8 = mask.contiguous().view(torch.uint8)
        o = torch.empty_like(q)
        lse = torch.empty((B * num_heads, 128), dtype=torch.float32, device=q.device)
        wsb = lib.usc_attn_ws_bytes(L, S, B, num_heads)
        ws = _ws(wsb, q.device)
        check(lib.usc_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(m8), L, S, B, num_heads, E, _ptr(o), _ptr(lse), _ptr(ws),
                               wsb, _stream()), "usc_attn_fwd")
        ctx.save_for_backward(q, k, v, m8, o, lse, ws)       # ws: its head keeps the packed mask for the backward
        ctx.num_heads = num_heads
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, m8, o, lse, ws = ctx.saved_tensors
        L, B, E = q.shape
        S = k.shape[0]
        do = do.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        wsb = lib.usc_attn_ws_bytes(L, S, B, ctx.num_heads)
        check(lib.usc_attn_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(m8), _ptr(o), _ptr(lse), _ptr(do), L, S, B, ctx.num_heads,
                               E, _ptr(dq), _ptr(dk), _ptr(dv), 1, _ptr(ws), wsb, _stream()), "usc_attn_bwd")
        return dq, dk, dv, None, None


def masked_cross_attention(q, k, v, mask_bsl, num_heads):
    """softmax(q k^T / 4 + mask) v per head for head dim 16: q [L,B,E], k/v [S,B,E] (f32, sequence-first),
    mask_bsl bool[B,S,L] with True = masked (the decoder's `batched_attn`, shared by all heads) -> [L,B,E]."""
    L, B, E = q.shape
    if E != 16 * num_heads or L > 128 or mask_bsl.dtype != torch.bool or tuple(mask_bsl.shape) != (B, k.shape[0], L):
        raise RuntimeError("masked_cross_attention: needs head dim 16, <= 128 queries and a bool[B,S,L] mask")
    return _MaskedCrossAttention.apply(q, k, v, mask_bsl, num_heads)


class _SelfAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, num_heads):
        L, B, E = q.shape
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        o = torch.empty_like(q)
        lse = torch.empty((B * num_heads, 128), dtype=torch.float32, device=q.device)
        check(lib.usc_self_attn_fwd(_ptr(q), _ptr(k), _ptr(v), L, B, num_heads, E, _ptr(o), _ptr(lse), _stream()),
              "usc_self_attn_fwd")
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.num_heads = num_heads
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        L, B, E = q.shape
        do = do.contiguous()
        d3 = torch.empty((3,) + tuple(q.shape), dtype=torch.float32, device=q.device)   # adjacent: the fused projection
        dq, dk, dv = d3[0], d3[1], d3[2]                                                # backward reads them as one table
        check(lib.usc_self_attn_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse), _ptr(do), L, B, ctx.num_heads, E,
                                    _ptr(dq), _ptr(dk), _ptr(dv), _stream()), "usc_self_attn_bwd")
        return dq, dk, dv, None


def self_attention(q, k, v, num_heads):
    """softmax(q k^T / 4) v per head for head dim 16, no mask: q, k, v f32[L,B,E] (sequence-first) with the same
    L <= 128 — the attention core of the decoder's SelfAttentionLayer (reference models/mask3d.py:491-545), one
    launch each way, bit-reproducible."""
    L, B, E = q.shape
    if E != 16 * num_heads or L > 128 or k.shape != q.shape or v.shape != q.shape:
        raise RuntimeError("self_attention: needs head dim 16, <= 128 queries and q, k, v of one shape")
    for t, name in ((q, "q"), (k, "k"), (v, "v")):
        if not t.is_cuda or t.dtype != torch.float32:
            raise RuntimeError(f"self_attention: {name} must be an f32 HIP tensor")
    return _SelfAttention.apply(q, k, v, num_heads)


def lsap_batch(cost: torch.Tensor):
    """scipy.optimize.linear_sum_assignment of every matrix of cost f32[P, nr, nc] on the device ->
    (row_ind i64[P, m], col_ind i64[P, m], status i32[P]) with m = min(nr, nc); ties resolve like scipy's
    (usc_lsap_batch).  status != 0 marks an infeasible problem (infinite / NaN costs; scipy raises ValueError there) —
    it stays on the device: the caller decides when to look."""
    require_device()
    _chk(cost, torch.float32, "cost")
    if cost.dim() != 3:
        raise RuntimeError("lsap_batch: cost must be f32 [n_problems, nr, nc]")
    P, nr, nc = cost.shape
    m = min(nr, nc)
    row = torch.empty((P, m), dtype=torch.int64, device=cost.device)
    col = torch.empty((P, m), dtype=torch.int64, device=cost.device)
    status = torch.empty(max(P, 1), dtype=torch.int32, device=cost.device)
    check(lib.usc_lsap_batch(_ptr(cost), P, nr, nc, _ptr(row), _ptr(col), _ptr(status), _stream()), "usc_lsap_batch")
    return row, col, status[:P]
