"""ctypes binding of libusc3d_hip.so — the flat C ABI declared in include/usc3d.h.

The library is the ONLY compute backend of this package: there is no CPU or
PyTorch fallback.  Loading fails loudly (ImportError) when the shared object has
not been built (`python -c "import __graft_entry__ as g; g.build()"` or
`make -C unscene3d_amd/csrc`), and every op raises RuntimeError when no HIP
device is present.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  It has to be in the process BEFORE this library
# pulls one in: with the order reversed, two runtimes coexist, this library sees the GPU and torch.cuda reports no
# device (found in round 2 by a test module that imported the package before torch).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("USC3D_LIB") or os.path.join(_HERE, "libusc3d_hip.so")   # override: developer ablation builds

_p = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_f64 = C.c_double
_f32 = C.c_float

class KMap(C.Structure):
    """usc_kmap (include/usc3d.h): one kernel map of the batch."""
    _fields_ = [("nbr", _p), ("perm", _p), ("tile_mask", _p), ("pair_in", _p), ("pair_out", _p), ("koff", _p),
                ("n_in", _i64), ("n_out", _i64), ("pair_capacity", _i64), ("K", _i32), ("reserved", _i32)]


class BNDesc(C.Structure):
    """usc_bn (include/usc3d.h)."""
    _fields_ = [("gamma", _p), ("beta", _p), ("running_mean", _p), ("running_var", _p), ("num_batches_tracked", _p),
                ("eps", _f32), ("momentum", _f32), ("c", _i32), ("training", _i32)]


_kp = C.POINTER(KMap)
_bp = C.POINTER(BNDesc)


class Step(C.Structure):
    """usc_step (include/usc3d.h): one step of a step program (usc_program_run)."""
    _fields_ = [("op", _i32), ("kind", _i32), ("cin", _i32), ("cout", _i32), ("relu", _i32),
                ("dx_accumulate", _i32), ("dW_accumulate", _i32), ("dbn_accumulate", _i32), ("defer_wgrad", _i32),
                ("accumulate", _i32),
                ("map", _p), ("bn", _p),          # (addresses of a KMap / BNDesc: C.addressof)
                ("x", _p), ("W", _p), ("residual", _p), ("y", _p), ("stats", _p), ("out", _p),
                ("dout", _p), ("dy", _p), ("dres", _p), ("dx", _p), ("dW", _p), ("dgamma", _p), ("dbeta", _p),
                ("a", _p), ("b", _p), ("dst", _p), ("dst2", _p), ("n", _i64), ("ca", _i32), ("cb", _i32)]


_sp = C.POINTER(Step)
STEP_UNIT_FWD, STEP_UNIT_BWD, STEP_CAT, STEP_SPLIT, STEP_ADD = 0, 1, 2, 3, 4

# name -> (restype, [argtypes])   — mirrors include/usc3d.h one to one
SIGNATURES = {
    "usc_last_error": (C.c_char_p, []),
    "usc_abi_version": (C.c_int, []),
    "usc_build_info": (C.c_char_p, []),
    "usc_device_count": (C.c_int, []),
    "usc_voxel_floor_f64": (C.c_int, [_p, _i64, _f64, _p, _p]),
    "usc_voxel_floor_f64_host": (C.c_int, [_p, _i64, C.c_double, _p]),
    "usc_unique_coords_host": (C.c_int, [_p, _i64, _i32, _p, _p, _p]),
    "usc_coordmap_capacity": (_i64, [_i64]),
    "usc_coordmap_ws_bytes": (_i64, [_i64]),
    "usc_coordmap_build": (C.c_int, [_p, _i64, _i32, _p, _p, _i64, _p, _p, _p, _p, _p, _i64, _p]),
    "usc_morton_cell_ids": (C.c_int, [_p, _i64, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    "usc_kernel_map_cube": (C.c_int, [_p, _i64, _i32, _i32, _p, _p, _i64, _p, _p]),
    "usc_kernel_map_down2": (C.c_int, [_p, _i64, _i32, _p, _p, _i64, _p, _p, _p]),
    "usc_rulebook_ws_bytes": (_i64, [_i64, _i64]),
    "usc_rulebook_compact": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _p, _i64, _p]),
    "usc_weight_transpose": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p]),
    "usc_spconv_plan": (C.c_int, [_i32, _i64, _i32, _i32, _i32]),
    "usc_spconv_gather_gemm_ws_bytes": (_i64, [_i64, _i32, _i32, _i32]),
    "usc_spconv_gather_gemm": (C.c_int, [_p, _i64, _i32, _p, _i32, _i32, _p, _i64, _p, _p, _i32, _i32, _p, _i64, _p]),
    "usc_rowsort_ws_bytes": (_i64, [_i32, _i64]),
    "usc_rowsort_build": (C.c_int, [_p, _i32, _i64, _p, _p, _p, _i64, _p]),
    "usc_spconv_sorted_ws_bytes": (_i64, [_i64, _i32, _i32, _i32]),
    "usc_spconv_sorted_gemm": (C.c_int, [_p, _i64, _i32, _p, _i32, _i32, _p, _p, _p, _i64, _p, _p, _i32, _i32, _p, _i64, _p]),
    "usc_spconv_sorted_gemm_ex": (C.c_int, [_p, _i64, _i32, _p, _i32, _i32, _p, _p, _p, _i64, _p, _p, _i32, _i32, _p, _i64, _p, _p]),
    "usc_group_reduce": (C.c_int, [_p, _i32, _i64, _i32, _p, _i32, _p, _p]),
    "usc_bn_tile_max_rows": (C.c_int64, []),
    "usc_bn_tile_ok": (C.c_int, [_i64, _i32]),
    "usc_bn_tile_ws_bytes": (C.c_int64, [_i32]),
    "usc_bn_tile_forward": (C.c_int, [_p, _i32, _p, _i64, _i32, _p, _p, _f32, _f32, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p,
                                      _p, _i64, _p]),
    "usc_bn_tile_backward": (C.c_int, [_p, _i32, _i32, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p,
                                       _i64, _p]),
    "usc_spconv_pairs_gemm": (C.c_int, [_p, _i32, _p, _i32, _i32, _p, _p, _p, _i64, _p, _p]),
    "usc_spconv_wgrad_ws_bytes": (_i64, [_i32, _i32, _i32]),
    "usc_spconv_wgrad": (C.c_int, [_p, _i32, _p, _i32, _i32, _p, _p, _p, _i64, _p, _i32, _p, _i64, _p]),
    "usc_spconv_wgrad_ws_bytes_rows": (_i64, [_i32, _i32, _i32, _i64]),
    "usc_step_size": (_i32, []),
    "usc_program_ws_bytes": (_i64, [_sp, _i32]),
    "usc_program_run": (C.c_int, [_sp, _i32, _i32, _p, _i64, _p]),
    "usc_spconv_wgrad_group_max": (_i32, []),
    "usc_spconv_wgrad_grid_limit": (None, [_i32]),
    "usc_spconv_wgrad_group_ok": (C.c_int, [_i32, _i32, _i32, _i32, _i64]),
    "usc_spconv_wgrad_group": (C.c_int, [_i32, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _i64, _i32, _p]),
    "usc_spconv_wgrad_table_ws_bytes": (_i64, [_i32, _i32, _i32]),
    "usc_spconv_wgrad_table": (C.c_int, [_p, _i32, _p, _i32, _p, _i32, _i64, _p, _i32, _p, _i64, _p]),
    "usc_set_side_stream": (C.c_int, [_p]),
    "usc_set_wgrad_lane": (C.c_int, [_p, _p, _i64, _i64]),
    "usc_wgrad_lane_join": (C.c_int, [_p]),
    "usc_wgrad_lane_hold": (C.c_int, [_i32, _i64, _i64, _p]),
    "usc_wgrad_lane_holding": (_i32, []),
    "usc_launch_stats_begin": (C.c_int, [_p, _i64, _p]),
    "usc_launch_stats_end": (_i64, [_p, _i64]),
    "usc_wall_clock_khz": (_i64, []),
    "usc_conv_ws_bytes": (_i64, [_kp, _i32, _i32, _i32]),
    "usc_unit_ws_bytes": (_i64, [_kp, _i32, _i32, _i32]),
    "usc_conv_forward": (C.c_int, [_kp, _i32, _p, _i32, _p, _i32, _p, _p, _p, _i64, _p]),
    "usc_conv_backward": (C.c_int, [_kp, _i32, _p, _i32, _p, _i32, _p, _p, _i32, _p, _i32, _p, _i64, _p]),
    "usc_bn_eval_stats": (C.c_int, [_p, _p, _p, _p, _f32, _i32, _p, _p, _p, _p, _p]),
    "usc_conv_bn_act_forward": (C.c_int, [_kp, _i32, _p, _i32, _p, _i32, _bp, _p, _i32, _p, _p, _p, _p, _i64, _p]),
    "usc_conv_bn_act_backward": (C.c_int, [_kp, _i32, _p, _i32, _p, _i32, _bp, _p, _p, _p, _p, _p, _p, _p, _i32, _p,
                                           _i32, _p, _p, _i32, _p, _i64, _p]),
    "usc_colstats_ws_bytes": (_i64, [_i64, _i32]),
    "usc_colstats": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _i64, _p]),
    "usc_bn_apply": (C.c_int, [_p, _p, _p, _p, _i32, _p, _i64, _i32, _p]),
    "usc_bn_backward_dx": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p]),
    "usc_bn_backward_reduce": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _i64, _p]),
    "usc_bn_forward_stats": (C.c_int, [_p, _i64, _i32, _p, _p, _f32, _f32, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "usc_relu_fwd": (C.c_int, [_p, _p, _i64, _p]),
    "usc_relu_bwd": (C.c_int, [_p, _p, _p, _i64, _p]),
    "usc_avgpool_down2": (C.c_int, [_p, _i32, _p, _i64, _p, _p]),
    "usc_avgpool_down2_ex": (C.c_int, [_p, _i32, _i32, _p, _p, _i64, _p, _p, _p]),
    "usc_gather_rows": (C.c_int, [_p, _i32, _p, _i64, _p, _p]),
    "usc_scatter_add_rows": (C.c_int, [_p, _i32, _p, _i64, _p, _p]),
    "usc_scatter_rows_unique": (C.c_int, [_p, _i32, _p, _i64, _p, _p]),
    "usc_scatter_rows_unique_add": (C.c_int, [_p, _i32, _p, _i64, _p, _p]),
    "usc_sample_keys_ws_bytes": (_i64, [_i32, _i32, _i32]),
    "usc_sample_keys": (C.c_int, [_p, _i32, _p, _i32, _p, _i32, _p, _i32, _i32, _p, _p, _p, _p, _p, _i64, _p]),
    "usc_segment_csr_ws_bytes": (_i64, [_i64, _i64]),
    "usc_segment_csr": (C.c_int, [_p, _i64, _i64, _p, _p, _p, _i64, _p]),
    "usc_segment_mean_fwd": (C.c_int, [_p, _i32, _p, _p, _i64, _p, _p]),
    "usc_segment_mean_bwd": (C.c_int, [_p, _i32, _p, _p, _i64, _p, _p]),
    "usc_segment_mean_nonzero": (C.c_int, [_p, _i32, _p, _p, _i64, _p, _p, _p]),
    "usc_segment_max_nonzero": (C.c_int, [_p, _i32, _p, _p, _i64, _p, _p, _p]),
    "usc_ncut_similarity": (C.c_int, [_p, _i64, _i32, _i32, _p, _p, _p]),
    "usc_ncut_similarity_masked": (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _p, _p]),
    "usc_ncut_normalize_mat": (C.c_int, [_p, _i64, _p, _i64, _p]),
    "usc_ncut_binarize": (C.c_int, [_p, _p, _i64, _f32, _f64, _p, _p, _p, _p]),
    "usc_ncut_fiedler_ws_bytes": (_i64, [_i64]),
    "usc_ncut_fiedler": (C.c_int, [_p, _p, _i64, _f64, _p, _p, _p, _i64, _p]),
    "usc_knn1": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _p]),
    "usc_cc_eps_init": (C.c_int, [_p, _i64, _p]),
    "usc_cc_eps_step": (C.c_int, [_p, _i64, _f32, _p, _p, _p, _p]),
    "usc_cc_eps_finish": (C.c_int, [_p, _i64, _p, _p, _p]),
    "usc_project_planes_fwd": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _i32, _i32] + [_p] * 10),
    "usc_project_planes_bwd": (C.c_int, [_p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p]),
    "usc_brick_mask_build": (C.c_int, [_p, _i64, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "usc_raycast_first_hit_map": (C.c_int, [_p, _p, _i64, _i64, _p, _p, _i32, _i32, _i32, _p, _p, _i32, _i32, _i32, _i32,
                                            _f32, _f32, _f32, _p, _p, _p]),
    "usc_raycast_first_hit_dense": (C.c_int, [_p, _i32, _i32, _i32, _i64, _p, _p, _i32, _i32, _i32, _i32, _f32, _f32,
                                              _f32, _p, _p, _p]),
    "usc_project_reduce": (C.c_int, [_p, _i32, _p, _p, _i64, _i32, _p, _p, _p]),
    "usc_project_predictions": (C.c_int, [_p, _i32, _p, _i64, _p, _p]),
    "usc_unproject_depth": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _p, _p]),
    "usc_spin": (C.c_int, [_i64, C.c_int, _p]),
    "usc_adamw_step": (C.c_int, [_p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _f32, _i64, _p]),
    "usc_elastic_displace": (C.c_int, [_p, _i32, _i64, _i32, _p, _i32, _i32, _i32, _p, _p, _p, _f64, _p, _p]),
    "usc_attn_ws_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "usc_attn_fwd": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _i64, _p]),
    "usc_attn_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _i64, _p]),
    "usc_lsap_batch": (C.c_int, [_p, _i32, _i32, _i32, _p, _p, _p, _p]),
    "usc_criterion_ws_bytes": (_i64, [_i32, _i32, _i32]),
    "usc_criterion_target_bits": (C.c_int, [_p, _i32, _i32, _p, _p, _p]),
    "usc_criterion_costs": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _i64, _i64, _i32, _p, _f32, _f32, _f32,
                                      _p, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "usc_criterion_losses": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p]),
    "usc_criterion_table": (C.c_int, [_p, _i32, _i32, _p, _p, _p]),
    "usc_criterion_backward": (C.c_int, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                         _i32, _i64, _i64, _p, _p]),
    "usc_self_attn_fwd": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p]),
    "usc_self_attn_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p]),
    "usc_linear_fwd": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _p, _p]),
    "usc_linear_bwd": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _i32, _p]),
    "usc_layernorm_fwd": (C.c_int, [_p, _p, _p, _i64, _i32, _f32, _p, _p, _p, _p]),
    "usc_add_layernorm_fwd": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _f32, _p, _p, _p, _p, _p]),
    "usc_linear_fwd_ex": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "usc_linear_fwd_pad": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _i32, _p]),
    "usc_linear_fwd_split": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    "usc_linear_bwd_ex": (C.c_int, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _i32, _p]),
    "usc_qkv_proj_bwd": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _p, _p, _p, _p, _p, _i32, _p]),
    "usc_linear_bwd_ex2": (C.c_int, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _p]),
    "usc_layernorm_bwd_ws_bytes": (_i64, [_i64, _i32]),
    "usc_layernorm_bwd": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _i32, _p, _i64, _p]),
    "usc_layernorm_bwd_ex": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _p, _i32, _p, _i64, _p]),
    "usc_affine_rows": (C.c_int, [_p, _i32, _i64, _i32, _p, _p, _p]),
    "usc_colsum_sequential": (C.c_int, [_p, _i64, _i32, _i32, _p, _p]),
    "usc_col_sum_ws_bytes": (_i64, [_i64, _i32]),
    "usc_col_sum": (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _i64, _p]),
    "usc_color_lut": (C.c_int, [_p, _i64, _i32, _p, _p, _i32, _p]),
    "usc_felz_face_normals": (C.c_int, [_p, _p, _i64, _p, _p]),
    "usc_felz_vertex_normals": (C.c_int, [_p, _p, _p, _i64, _p, _p]),
    "usc_felz_edge_weights": (C.c_int, [_p, _p, _p, _p, _i64, _p, _p, _p, _p]),
    "usc_felz_merge_host": (C.c_int, [_p, _p, _p, _i64, _i32, _f32, _i32, _p]),
    "usc_furthest_point_sampling": (C.c_int, [_p, _i32, _i32, _i32, _p, _p, _p]),
    "usc_fourier_posenc": (C.c_int, [_p, _i64, _p, _p, _p, _i32, _p, _p]),
}


class LaunchStat(C.Structure):
    """Mirror of usc_launch_stat (include/usc3d.h)."""
    _fields_ = [("n_out", _i64), ("cin", _i32), ("cout", _i32), ("K", _i32), ("nb", _i32)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(`python -c 'import __graft_entry__ as g; g.build()'`). "
            "unscene3d_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
if lib.usc_step_size() != C.sizeof(Step):
    raise ImportError(f"unscene3d_amd._lib.Step ({C.sizeof(Step)} bytes) does not mirror usc_step "
                      f"({lib.usc_step_size()} bytes): rebuild the library or fix the binding")


def last_error() -> str:
    return lib.usc_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    """Turn a usc_status into a Python RuntimeError (reference: TORCH_CHECK ->
    RuntimeError, utils/cuda_utils/cuda_utils.cpp:4-6)."""
    if rc != 0:
        raise RuntimeError(f"{what or 'libusc3d_hip'} failed (status {rc}): {last_error()}")


def require_device() -> None:
    n = lib.usc_device_count()
    if n <= 0:
        raise RuntimeError(
            "unscene3d_amd: no HIP device visible — the MI355X kernels cannot run and there is no CPU fallback"
        )
