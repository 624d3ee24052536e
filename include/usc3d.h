/*
 * usc3d.h — flat C ABI of libusc3d_hip.so: the MI355X (gfx950) hot path of
 * UnScene3D's self-training step and pseudo-mask generator.
 *
 * Every entry point takes raw DEVICE pointers (caller-owned, contiguous,
 * row-major), plain sizes and an explicit stream (hipStream_t passed as
 * void*).  No entry point allocates, frees, synchronises or keeps a pointer
 * beyond the call; scratch memory is passed in by the caller (sizes via the
 * *_ws_bytes helpers).  Every function returns 0 on success or a negative
 * usc_status; usc_last_error() gives the message (the Python wrapper raises
 * RuntimeError — never exit()/abort(), cf. reference cuda_utils.cpp:4-6 vs
 * pointnet2 cuda_utils.h:32-41).
 *
 * Each declaration cites the reference interface it replaces
 * (paths relative to RozDavid/UnScene3D @ 2024_10_08).
 * [ME] = MinkowskiEngine 0.5.4, an un-vendored dependency of the reference
 * (conf/unscene3d_requirements.txt:50); semantics restated in oracle/.
 */
#ifndef USC3D_H
#define USC3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* usc_stream_t; /* hipStream_t */

enum usc_status {
  USC_OK = 0,
  USC_ERR_ARG = -1,     /* bad argument (null pointer, unsupported shape) */
  USC_ERR_LAUNCH = -2,  /* HIP launch/runtime error */
  USC_ERR_RANGE = -3    /* coordinate outside the packable range */
};

/* Message for the last failing call on this thread ("" if none). */
const char* usc_last_error(void);
/* Library/ABI version (bumped on any signature change). */
int usc_abi_version(void);
/* What the loaded library was built from: "arch=gfx950; HIP version: ...; built <UTC time>; sources sha256 <16 hex>" (the
 * hash covers every .hip / .cpp / .h the objects were compiled from).  __graft_entry__.build() and smoke() print it. */
const char* usc_build_info(void);
/* Number of visible HIP devices, or a negative usc_status. */
int usc_device_count(void);

/* ------------------------------------------------------------------------
 * V1/V3/R1  coordinate maps  — replaces [ME] ME.utils.sparse_quantize
 * (datasets/utils.py:403-408, pseudo_masks/datasets/voxelizer.py:142) and the
 * coordinate-manager insert behind ME.SparseTensor(...) (trainer/trainer.py:115)
 * and the stride-2 maps implied by MinkowskiConvolution(k=2,s=2)
 * (models/res16unet.py:51-116).
 * ---------------------------------------------------------------------- */

/* coords_out[i,:] = floor(xyz[i,:] / voxel_size) in float64 (== np.floor(x/v),
 * datasets/utils.py:403), cast to int32.  xyz: f64[n,3]. */
int usc_voxel_floor_f64(const double* xyz, int64_t n, double voxel_size,
                        int32_t* coords_out, usc_stream_t s);

/* HIP-FREE, fork-safe host forms (plain C++ over HOST pointers; no HIP call, no global state): the reference voxelises
 * in forked DataLoader workers on the CPU (datasets/utils.py:403-414, conf/data/indoor.yaml:24), where the parent's HIP
 * runtime must not be touched.  usc_voxel_floor_f64_host = usc_voxel_floor_f64; usc_unique_coords_host = the unique_idx /
 * inverse pair of usc_coordmap_build(quant = 1) on coords i32[n,d], d = 3 (x,y,z) or 4 (b,x,y,z): first-occurrence rows
 * in ascending order — ME.utils.sparse_quantize(return_index=True, return_inverse=True).  Bit-equal to the device path. */
int usc_voxel_floor_f64_host(const double* xyz, int64_t n, double voxel_size,
                             int32_t* coords_out);
int usc_unique_coords_host(const int32_t* coords, int64_t n, int32_t d,
                           int64_t* unique_idx, int64_t* inverse,
                           int64_t* n_out);

/* Hash-table capacity (slots, power of two) required for n coordinates. */
int64_t usc_coordmap_capacity(int64_t n);
/* Scratch bytes for usc_coordmap_build on n rows. */
int64_t usc_coordmap_ws_bytes(int64_t n);

/* Build a coordinate map (open-addressing hash on the packed 64-bit key of
 * (b, x, y, z)) over coords i32[n,4], optionally quantising x,y,z to
 * multiples of `quant` first (quant<=1: none; quant = 2*tensor_stride builds
 * the next-coarser map: c -> floor(c/quant)*quant).
 * Outputs, all caller-allocated:
 *   table_keys u64[cap], table_vals i32[cap]  — persistent map: key -> row
 *   unique_idx i64[n]   first n_out entries valid: the FIRST-OCCURRENCE input
 *                       row of each distinct coordinate, ascending
 *                       (== unique_map of sparse_quantize)
 *   inverse    i64[n]   row of the distinct coordinate each input row maps to
 *                       (== inverse_map)
 *   out_coords i32[n,4] first n_out rows valid: the distinct (quantised)
 *                       coordinates in map-row order (may be NULL)
 *   n_out      i64[1]   device scalar
 */
int usc_coordmap_build(const int32_t* coords, int64_t n, int32_t quant,
                       uint64_t* table_keys, int32_t* table_vals, int64_t cap,
                       int64_t* unique_idx, int64_t* inverse,
                       int32_t* out_coords, int64_t* n_out,
                       void* ws, int64_t ws_bytes, usc_stream_t s);

/* Spatial (z-order) cell id of every row: Morton code of ((c - lo) >> shift) with
 * bits_per_axis bits per axis, batch index above it.  Feeding these ids to
 * usc_segment_csr gives a stable, cache-friendly row permutation (rows of one
 * 2^shift-voxel cell become contiguous) — an MI355X-side optimisation of the
 * collate step (datasets/utils.py:412-417 gathers by unique_map; any consistent
 * row order is valid downstream); it keeps gathers of neighbouring rows inside
 * one XCD's L2. */
int usc_morton_cell_ids(const int32_t* coords, int64_t n, int32_t shift,
                        int32_t lo_x, int32_t lo_y, int32_t lo_z,
                        int32_t bits_per_axis, int64_t* ids, usc_stream_t s);

/* ------------------------------------------------------------------------
 * R2  kernel maps ("rulebooks") — replaces [ME] the kernel-map construction
 * cached by the CoordinateManager for MinkowskiConvolution /
 * MinkowskiConvolutionTranspose / MinkowskiAvgPooling
 * (models/modules/common.py:125-188, models/mask3d.py:131).
 * ---------------------------------------------------------------------- */

/* Dense neighbour table for a stride-1 k^3 HYPER_CUBE kernel on one map:
 *   nbr[k*n + o] = row i with coord_i == coord_o + off_k*tensor_stride, else -1
 * offsets: ksize=3 -> {-1,0,1}^3, x fastest (k = dx+1 + 3(dy+1) + 9(dz+1)).
 * coords i32[n,4] are the map's coordinates; table_* its hash table. */
int usc_kernel_map_cube(const int32_t* coords, int64_t n, int32_t tensor_stride,
                        int32_t ksize, const uint64_t* table_keys,
                        const int32_t* table_vals, int64_t cap, int32_t* nbr,
                        usc_stream_t s);

/* Child table of a k=2,s=2 kernel between a fine map (n_fine rows, tensor
 * stride ts) and its coarse parent (n_coarse rows):
 *   nbr2[k*n_coarse + p] = fine row whose parent is p and whose offset index is
 *       k = ox + 2*oy + 4*oz, o = (c_fine - c_parent)/ts in {0,1}^3, else -1
 *   kidx[i] = k of fine row i (u8)
 * parent i64[n_fine] is the `inverse` returned by usc_coordmap_build(quant=2ts).
 * nbr2 must be pre-filled with -1 by the caller. */
int usc_kernel_map_down2(const int32_t* fine_coords, int64_t n_fine,
                         int32_t tensor_stride, const int64_t* parent,
                         const int32_t* coarse_coords, int64_t n_coarse,
                         int32_t* nbr2, uint8_t* kidx, usc_stream_t s);

/* Scratch bytes for usc_rulebook_compact on a [K, n_out] table. */
int64_t usc_rulebook_ws_bytes(int64_t K, int64_t n_out);
/* Compact a dense neighbour table into per-offset pair lists, ordered by
 * (k, out row):  for p in [koff[k], koff[k+1]): (in_idx[p], out_idx[p]).
 * in_idx/out_idx i32[K*n_out] capacity, koff i64[K+1] (device). */
int usc_rulebook_compact(const int32_t* nbr, int64_t K, int64_t n_out,
                         int32_t* in_idx, int32_t* out_idx, int64_t* koff,
                         void* ws, int64_t ws_bytes, usc_stream_t s);

/* ------------------------------------------------------------------------
 * C  sparse convolution — replaces [ME] MinkowskiConvolution /
 * MinkowskiConvolutionTranspose forward + backward
 * (models/modules/common.py:146,179; instances models/res16unet.py:39-221,
 * models/modules/resnet_block.py:24-43, models/mask3d.py:60-62).
 * fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32.
 * ---------------------------------------------------------------------- */

/* out[k][co][ci] = W[mirror ? K-1-k : k][ci][co]  — weights for the dgrad
 * kernels (mirror=1 for a stride-1 odd kernel whose map is its own transpose). */
int usc_weight_transpose(const float* W, int32_t K, int32_t cin, int32_t cout,
                         int32_t mirror, float* out, usc_stream_t s);

/* Output-stationary implicit GEMM over a dense neighbour table:
 *   out[o,:] = sum_k  in[nbr[k*n_out+o], :] @ W[k]  (+ bias)
 * in f32[n_in,cin], W f32[K,cin,cout].  nbr==NULL means K==1 identity
 * (1x1 conv / linear).  bias f32[cout] or NULL.  accumulate=1 adds into out.
 * Small maps (coarse U-Net levels) split the K offsets over extra workgroups and
 * reduce the partial sums in a fixed order through `ws`
 * (usc_spconv_gather_gemm_ws_bytes; 0 bytes when no split is planned).
 * w_transposed=1 (tile-compacted plan, usc_spconv_plan bit 12; also K = 1 on identity rows with cin, cout multiples of
 * 32 — a linear layer's [cout][cin] weight read in place): W is the forward
 * conv's f32[K,cout,cin] and W'[k][c][n] = W[K-1-k][n][c] is used (stride-1 dgrad).
 * Covers: k3/s1 conv fwd and dgrad (W = usc_weight_transpose(mirror=1), or w_transposed),
 * k2/s2 conv fwd (nbr = child table), conv-transpose dgrad. */
/* Launch plan chosen for a shape (for profiling labels): returns
 * NB | aligned<<8 | G<<16 where NB = 32-column blocks per wave, aligned = fast
 * path, G = offset groups.  kind 0: gather_gemm (n = n_out), 1: pairs_gemm
 * (n = P_capacity), 2: wgrad (bit 13 set: the all-input-tiles kernel, CT = code>>16). */
int usc_spconv_plan(int32_t kind, int64_t n, int32_t cin, int32_t cout,
                    int32_t K);
/* Mask-sorted form of the same convolution (all map sizes, cin and cout multiples
 * of 32).  usc_rowsort_build groups the OUTPUT rows of a neighbour table by their
 * neighbour bitmask (one bucket pass on up to 12 informative offset bits):
 *   perm      i32[n_out]            output rows, similar masks adjacent
 *   tile_mask u32[ceil(n_out/32)]   OR of the masks of each run of 32 rows of perm
 * so that a 32-row matrix-core tile skips every offset none of its rows has.
 * usc_spconv_sorted_gemm computes usc_spconv_gather_gemm's result; its output
 * bits do not depend on perm (each output element is reduced by one lane over
 * k ascending, channel ascending); weights are staged through LDS per workgroup.
 * One perm serves the forward conv and, for a stride-1 map, its input gradient:
 * w_transposed=1 takes the FORWARD weights f32[K, cout, cin] and uses
 * W'[k][c][n] = W[K-1-k][n][c] while staging them (no usc_weight_transpose pass).
 * Replaces the same MinkowskiEngine calls as usc_spconv_gather_gemm
 * (models/res16unet.py:224-297; ME 0.5.4 src/convolution_kernel.cu, un-vendored). */
int64_t usc_rowsort_ws_bytes(int32_t K, int64_t n_out);
int usc_rowsort_build(const int32_t* nbr, int32_t K, int64_t n_out,
                      int32_t* perm, uint32_t* tile_mask, void* ws,
                      int64_t ws_bytes, usc_stream_t s);
int64_t usc_spconv_sorted_ws_bytes(int64_t n_out, int32_t cin, int32_t cout,
                                   int32_t K);
int usc_spconv_sorted_gemm(const float* in, int64_t n_in, int32_t cin,
                           const float* W, int32_t K, int32_t cout,
                           const int32_t* nbr, const int32_t* perm,
                           const uint32_t* tile_mask, int64_t n_out,
                           const float* bias, float* out, int32_t accumulate,
                           int32_t w_transposed, void* ws, int64_t ws_bytes,
                           usc_stream_t s);
/* The same with the slice reduction left to the caller: when the launch plan splits the K offsets over G > 1 partial-sum
 * slices and `slices_left` is not NULL (and there is no bias), *slices_left = G, the slices f32[G][n_out][cout] stay at
 * the start of `ws`, and `out` / `accumulate` are NOT applied — the caller sums the slices in slice order, usually fused
 * with what follows the convolution (usc_bn_tile_forward / usc_bn_tile_backward below; usc_group_reduce is the plain
 * form).  *slices_left = 0: the call did everything, as usc_spconv_sorted_gemm.  Replaces the same MinkowskiEngine
 * calls (models/modules/common.py:125-188). */
int usc_spconv_sorted_gemm_ex(const float* in, int64_t n_in, int32_t cin,
                              const float* W, int32_t K, int32_t cout,
                              const int32_t* nbr, const int32_t* perm,
                              const uint32_t* tile_mask, int64_t n_out,
                              const float* bias, float* out, int32_t accumulate,
                              int32_t w_transposed, void* ws, int64_t ws_bytes,
                              int32_t* slices_left, usc_stream_t s);
/* out[n,c] = (accumulate ? out : 0) + bias + sum_g partial[g][n][c], g ascending (the reduction the split-K launches
 * end with; c a multiple of 4). */
int usc_group_reduce(const float* partial, int32_t G, int64_t n, int32_t c,
                     const float* bias, int32_t accumulate, float* out,
                     usc_stream_t s);

int64_t usc_spconv_gather_gemm_ws_bytes(int64_t n_out, int32_t cin,
                                        int32_t cout, int32_t K);
int usc_spconv_gather_gemm(const float* in, int64_t n_in, int32_t cin,
                           const float* W, int32_t K, int32_t cout,
                           const int32_t* nbr, int64_t n_out,
                           const float* bias, float* out, int32_t accumulate, int32_t w_transposed,
                           void* ws, int64_t ws_bytes, usc_stream_t s);

/* One-parent form (transposed-conv forward, strided-conv dgrad):
 *   out[rows_out[p],:] = in[rows_in[p],:] @ W[k]   for p in [koff[k], koff[k+1])
 * driven by the per-offset pair lists of usc_rulebook_compact on the child
 * table (every output row appears in exactly one pair).  koff i64[K+1] device,
 * P_capacity = upper bound of koff[K] known to the host. */
int usc_spconv_pairs_gemm(const float* in, int32_t cin, const float* W,
                          int32_t K, int32_t cout, const int32_t* rows_in,
                          const int32_t* rows_out, const int64_t* koff,
                          int64_t P_capacity, float* out, usc_stream_t s);

/* Scratch bytes for usc_spconv_wgrad. */
int64_t usc_spconv_wgrad_ws_bytes(int32_t K, int32_t cin, int32_t cout);
/* Weight gradient:  dW[k] = sum_{p in list k}  a[a_idx[p],:]^T  b[b_idx[p],:]
 * a f32[*,cin], b f32[*,cout], dW f32[K,cin,cout]; pair lists as produced by
 * usc_rulebook_compact (a_idx/b_idx i32, koff i64[K+1] device).
 * a_idx==NULL && K==1: identity pairs over n_rows (1x1 conv). Deterministic
 * (fixed split + ordered reduction, no float atomics).  accumulate=1 adds into
 * dW (a parameter's existing gradient buffer) instead of overwriting it. */
int usc_spconv_wgrad(const float* a, int32_t cin, const float* b, int32_t cout,
                     int32_t K, const int32_t* a_idx, const int32_t* b_idx,
                     const int64_t* koff, int64_t n_rows, float* dW, int32_t accumulate, void* ws,
                     int64_t ws_bytes, usc_stream_t s);
/* Exact scratch need of usc_spconv_wgrad for pair lists of capacity n_rows (the bound
 * above assumes the maximum split count). */
int64_t usc_spconv_wgrad_ws_bytes_rows(int32_t K, int32_t cin, int32_t cout,
                                       int64_t n_rows);
/* Weight gradient in TABLE form for few input channels (cin <= 4, cout = 32: the 3 -> 32 stem, reference
 * models/res16unet.py conv0p1s1):  dW[k][c][n] = sum_o in[nbr[k][o]][c] dy[o][n]  with the forward conv's neighbour
 * table nbr i32[K, n_out].  32 / cin offsets share one MFMA tile (row m = offset, channel), 3 groups instead of 27
 * per-offset passes with 3 useful tile rows each. */
int64_t usc_spconv_wgrad_table_ws_bytes(int32_t K, int32_t cin, int32_t cout);
int usc_spconv_wgrad_table(const float* in, int32_t cin, const float* dy, int32_t cout, const int32_t* nbr, int32_t K,
                           int64_t n_out, float* dW, int32_t accumulate, void* ws, int64_t ws_bytes, usc_stream_t s);

/* GROUPED weight gradient: R problems of one shape on one kernel map — the conv1 / conv2 weight gradients of a level's
 * residual blocks (reference models/modules/resnet_block.py:48-64 builds them, autograd differentiates them one by
 * one; models/res16unet.py LAYERS = 2..6 same-shape blocks per level) — in ONE grid: dW[r][k] (+)= sum_p a[r][a_idx[p]]^T
 * b[r][b_idx[p]].  a / b / dW: HOST arrays of R device pointers (copied into the launch; R <= usc_spconv_wgrad_group_max()).
 * No pair split and no reduction launch: every (problem, offset, channel tile) workgroup walks its whole pair list and
 * writes (accumulate = 1: adds into) dW[r][k] itself, in a fixed order.  usc_spconv_wgrad_group_ok: whether the grouped
 * form applies (channel pair covered, >= 256 workgroups, <= 2 048 pairs per offset: the coarse levels). */
/* > 0: usc_spconv_wgrad launches its all-input-tiles kernel with at most that many workgroups along x, each walking
 * several (offset, split) work items ("background form": the launch leaves wave slots and matrix-core time on every CU
 * to the kernels of another stream).  0 restores the plain launches.  Process-wide; set by the weight-gradient lane
 * around its own launches.  Same arithmetic, same results. */
void usc_spconv_wgrad_grid_limit(int32_t max_workgroups);
int32_t usc_spconv_wgrad_group_max(void);
int usc_spconv_wgrad_group_ok(int32_t R, int32_t cin, int32_t cout, int32_t K, int64_t n_rows);
int usc_spconv_wgrad_group(int32_t R, const float* const* a, const float* const* b, float* const* dW, int32_t cin,
                           int32_t cout, int32_t K, const int32_t* a_idx, const int32_t* b_idx, const int64_t* koff,
                           int64_t n_rows, int32_t accumulate, usc_stream_t s);

/* ------------------------------------------------------------------------
 * Native issue path: whole convolutions and "conv -> batch norm (+ residual)
 * (+ ReLU)" units behind ONE call each way — the operator granularity of
 * models/modules/resnet_block.py:48-64 (conv1/norm1/relu, conv2/norm2,
 * `out += residual`, relu) and of the stem / strided / transposed stages of
 * models/res16unet.py:231-297.  Same kernels and kernel choice as the
 * per-operator entry points above; the point is that the host issues 3-6
 * launches per call instead of one per call from the interpreter.
 *
 * usc_kmap describes one kernel map of the batch (built by usc_kernel_map_cube /
 * usc_kernel_map_down2, usc_rowsort_build, usc_rulebook_compact); all pointers
 * are device pointers owned by the caller's coordinate manager.
 *   kind USC_CONV_SAME : k^3 stride-1 conv on one map (n_in == n_out), or a 1x1
 *                        conv when nbr == NULL and K == 1
 *        USC_CONV_DOWN : k=2,s=2 conv, fine map (n_in rows) -> coarse map
 *                        (n_out rows); nbr = child table [8][n_out]
 *        USC_CONV_UP   : the transposed conv on the SAME map description:
 *                        input = coarse rows (n_out), output = fine rows (n_in)
 * ---------------------------------------------------------------------- */
typedef struct usc_kmap {
  const int32_t* nbr;        /* i32[K][n_out]; NULL: identity (1x1 conv) */
  const int32_t* perm;       /* usc_rowsort_build of nbr, or NULL */
  const uint32_t* tile_mask; /* usc_rowsort_build of nbr, or NULL */
  const int32_t* pair_in;    /* usc_rulebook_compact of nbr: rows of the n_in side */
  const int32_t* pair_out;   /*                              rows of the n_out side */
  const int64_t* koff;       /* i64[K+1] (device) */
  int64_t n_in, n_out;
  int64_t pair_capacity;     /* allocated length of pair_in / pair_out */
  int32_t K;
  int32_t reserved;
} usc_kmap;

enum usc_conv_kind { USC_CONV_SAME = 0, USC_CONV_DOWN = 1, USC_CONV_UP = 2 };

/* MinkowskiBatchNorm == BatchNorm1d over rows (models/modules/common.py:22).
 * training != 0: batch statistics (+ running-stat update when running_* != NULL,
 * num_batches_tracked incremented when != NULL); else running statistics. */
typedef struct usc_bn {
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  int64_t* num_batches_tracked;
  float eps, momentum;
  int32_t c;
  int32_t training;
} usc_bn;

/* Optional: a second stream of the CALLER (NULL switches it off again; per device).
 * usc_conv_backward / usc_conv_bn_act_backward then fork the weight gradient of
 * maps with <= 24 576 rows onto it and join before returning the stream to the
 * caller's order: on the coarse U-Net levels the input-gradient and weight-gradient
 * launches are latency-bound and run side by side.  Results are unchanged. */
int usc_set_side_stream(usc_stream_t side);
/* Optional: the weight-gradient LANE, a second stream of the caller with its own scratch (NULL lane: off; per device).
 * usc_conv_backward / usc_conv_bn_act_backward then queue the weight gradient of a map with <= max_rows rows (only
 * the accumulate-into-a-gradient-buffer form) on the lane behind one event of the caller's stream and return WITHOUT
 * waiting for it; the input-gradient chain of the backward pass is not held up, the latency-bound weight-gradient
 * launches of the coarse U-Net levels run beside it.  The caller keeps x, dy and dW alive and untouched until it has
 * called usc_wgrad_lane_join(s) — `s` then waits for everything queued on the lane so far (end of the backward pass,
 * and before a gradient bucket is handed to a collective).  Results are unchanged (same kernels, same order per
 * gradient buffer). */
int usc_set_wgrad_lane(usc_stream_t lane, void* lane_ws, int64_t lane_ws_bytes, int64_t max_rows);
int usc_wgrad_lane_join(usc_stream_t s);
/* Scheduling of the lane against the caller's chain (step program of the backbone's backward pass, which visits the
 * U-Net stages fine -> coarse -> fine, models/res16unet.py:224-297 in reverse).  mode 1: from now on the weight gradient
 * of a map with >= hold_min_rows rows is only NOTED; the first backward call on a map with <= release_max_rows rows —
 * a stage whose launches cannot fill 256 CUs — puts everything noted on the lane behind one event of `s` and ends the
 * hold (later calls queue at once, as without a hold).  The throughput-bound fine-level weight gradients then run
 * beside the latency-bound coarse-level chain instead of beside the fine-level input gradients they compete with for
 * matrix-core issue slots.  mode 0: release what is still noted and end the hold (also done by usc_wgrad_lane_join).
 * mode -1: forget what was noted (the caller's pass failed; the noted pointers are stale).  The caller keeps every
 * noted x / dy / dW alive until the join, as for a queued one.  Results are unchanged. */
int usc_wgrad_lane_hold(int32_t mode, int64_t hold_min_rows, int64_t release_max_rows, usc_stream_t s);
/* 1 while a hold is in force on the current device (nothing released yet), else 0. */
int32_t usc_wgrad_lane_holding(void);
/* Launch statistics of the tile-compacted kernel (the dominant kernel of the step, models/modules/common.py:125-188),
 * taken by the kernel itself while a real step runs — every stream, graph and lane as they are — so that the bench can
 * report the kernel's rate inside the step it times (`roofline.frac_in_step`) next to its rate alone.  Between begin and
 * end every such launch gets the next slot of `ring_dev` (device memory, slots x 4 x u64, initialised on `s` by begin):
 * {earliest workgroup start, latest workgroup end (wall-clock ticks, usc_wall_clock_khz), real (in, out) pairs of the
 * launch, 0}; launches beyond `slots` are not recorded.  usc_launch_stats_end stops the recording and copies the shapes
 * of the recorded launches, in launch order, into host_out (<= max_out entries); it returns their number.  The caller
 * synchronises before reading the ring.  Cost while recording: two atomics per workgroup. */
typedef struct usc_launch_stat { int64_t n_out; int32_t cin, cout, K, nb; } usc_launch_stat;
int usc_launch_stats_begin(void* ring_dev, int64_t slots, usc_stream_t s);
int64_t usc_launch_stats_end(usc_launch_stat* host_out, int64_t max_out);
int64_t usc_wall_clock_khz(void);
/* Scratch bytes covering forward AND backward of one convolution / one unit. */
int64_t usc_conv_ws_bytes(const usc_kmap* m, int32_t kind, int32_t cin,
                          int32_t cout);
int64_t usc_unit_ws_bytes(const usc_kmap* m, int32_t kind, int32_t cin,
                          int32_t cout);
/* y = conv(x; W) (+ bias; bias only for SAME/DOWN).  W f32[K,cin,cout]. */
int usc_conv_forward(const usc_kmap* m, int32_t kind, const float* x,
                     int32_t cin, const float* W, int32_t cout,
                     const float* bias, float* y, void* ws, int64_t ws_bytes,
                     usc_stream_t s);
/* dx (skipped when NULL; dx_accumulate adds into dx — gather forms only) and
 * dW (skipped when NULL; dW_accumulate adds into a gradient buffer). */
int usc_conv_backward(const usc_kmap* m, int32_t kind, const float* x,
                      int32_t cin, const float* W, int32_t cout,
                      const float* dy, float* dx, int32_t dx_accumulate,
                      float* dW, int32_t dW_accumulate, void* ws,
                      int64_t ws_bytes, usc_stream_t s);
/* mean/invstd/scale/shift of an eval-mode batch norm from its running statistics. */
int usc_bn_eval_stats(const float* gamma, const float* beta,
                      const float* running_mean, const float* running_var,
                      float eps, int32_t c, float* mean, float* invstd,
                      float* scale, float* shift, usc_stream_t s);
/* out = [relu]( BN(conv(x; W)) (+ residual) ).  Saved for the backward pass:
 * y = conv output f32[n,cout], stats f32[4][cout] = mean | invstd | scale | shift,
 * out (its sign is the ReLU mask). */
int usc_conv_bn_act_forward(const usc_kmap* m, int32_t kind, const float* x,
                            int32_t cin, const float* W, int32_t cout,
                            const usc_bn* bn, const float* residual,
                            int32_t relu, float* y, float* stats, float* out,
                            void* ws, int64_t ws_bytes, usc_stream_t s);
/* Backward of the unit.  out_relu = the forward output when relu was applied, else
 * NULL.  dy f32[n,cout] receives d(conv output) (scratch the caller may drop);
 * dres (or NULL) receives the residual branch's gradient (= masked dout).
 * dgamma/dbeta: written, or added into when dbn_accumulate. */
int usc_conv_bn_act_backward(const usc_kmap* m, int32_t kind, const float* x,
                             int32_t cin, const float* W, int32_t cout,
                             const usc_bn* bn, const float* y,
                             const float* stats, const float* out_relu,
                             const float* dout, float* dy, float* dres,
                             float* dx, int32_t dx_accumulate, float* dW,
                             int32_t dW_accumulate, float* dgamma,
                             float* dbeta, int32_t dbn_accumulate, void* ws,
                             int64_t ws_bytes, usc_stream_t s);

/* ------------------------------------------------------------------------
 * Step programs: the issue loop of a whole network stage behind ONE call.
 * The reference walks Res16UNetBase.forward (models/res16unet.py:224-297) module by module from the interpreter, and
 * autograd walks it back; here the caller describes the same walk once as an array of steps — conv+BN units, the
 * channel concatenation of a skip connection (`me.cat`, :259-289), and for the way back the units' backward calls, the
 * column split of a concatenation's gradient and the fan-in adds of tensors with several consumers — and
 * usc_program_run launches steps [begin, end) in order on the stream.  Every pointer is a device pointer owned by the
 * caller (activations, gradients and scratch live in the caller's arenas for as long as the steps need them); the
 * library allocates nothing and keeps nothing: deferred weight gradients (defer_wgrad != 0, see
 * usc_spconv_wgrad_group) are queued inside one call only and flushed before it returns.
 *   USC_STEP_UNIT_FWD : usc_conv_bn_act_forward(map, kind, x, cin, W, cout, bn, residual, relu, y, stats, out)
 *   USC_STEP_UNIT_BWD : usc_conv_bn_act_backward(map, kind, x, cin, W, cout, bn, y, stats, out (NULL: no ReLU),
 *                       dout, dy, dres, dx, dx_accumulate, dW, dW_accumulate, dgamma, dbeta, dbn_accumulate)
 *   USC_STEP_CAT      : dst[n, ca + cb] = [a[n, ca] | b[n, cb]]
 *   USC_STEP_SPLIT    : dst[n, ca] = a[:, :ca];  dst2[n, cb] (+= when accumulate) a[:, ca:ca + cb]   (a is [n, ca + cb])
 *   USC_STEP_ADD      : dst[0 .. n) += a[0 .. n)   (n counts floats)
 * ---------------------------------------------------------------------- */
enum usc_step_op { USC_STEP_UNIT_FWD = 0, USC_STEP_UNIT_BWD = 1, USC_STEP_CAT = 2, USC_STEP_SPLIT = 3, USC_STEP_ADD = 4 };
typedef struct usc_step {
  int32_t op, kind, cin, cout, relu;
  int32_t dx_accumulate, dW_accumulate, dbn_accumulate, defer_wgrad, accumulate;
  const usc_kmap* map;
  const usc_bn* bn;
  const float* x;
  const float* W;
  const float* residual;
  float* y;
  float* stats;
  float* out;
  const float* dout;
  float* dy;
  float* dres;
  float* dx;
  float* dW;
  float* dgamma;
  float* dbeta;
  const float* a;
  const float* b;
  float* dst;
  float* dst2;
  int64_t n;
  int32_t ca, cb;
} usc_step;
/* sizeof(usc_step) as the library was compiled: a binding checks its own mirror of the structure against it. */
int32_t usc_step_size(void);
/* Scratch bytes that cover every step of the program (the steps run one after the other on one stream). */
int64_t usc_program_ws_bytes(const usc_step* steps, int32_t n_steps);
int usc_program_run(const usc_step* steps, int32_t begin, int32_t end, void* ws, int64_t ws_bytes, usc_stream_t s);

/* ------------------------------------------------------------------------
 * B  row-wise batch norm / ReLU / residual — replaces [ME] MinkowskiBatchNorm
 * (= BatchNorm1d over feature rows), MinkowskiReLU and `out += residual`
 * (models/modules/common.py:22, models/modules/resnet_block.py:48-64).
 * ---------------------------------------------------------------------- */

int64_t usc_colstats_ws_bytes(int64_t n, int32_t c);
/* Per-channel sums over rows: sum1[c] = sum_i x[i,c], sum2[c] = sum_i x[i,c]*y[i,c]
 * (y==NULL -> x*x).  Deterministic two-stage reduction.  f32 in, f64 partials. */
int usc_colstats(const float* x, const float* y, int64_t n, int32_t c,
                 double* sum1, double* sum2, void* ws, int64_t ws_bytes,
                 usc_stream_t s);
/* BatchNorm1d training statistics in one pass (f64 accumulation) + finalisation:
 * mean, invstd = 1/sqrt(var_biased+eps), scale = gamma*invstd, shift = beta-mean*scale,
 * and the running-stat update running = (1-m)*running + m*{mean, var_unbiased}
 * (running_* may both be NULL).  ws: usc_colstats_ws_bytes(n, c).  num_batches_tracked (i64[1]
 * or NULL) is incremented by the same launch. */
int usc_bn_forward_stats(const float* x, int64_t n, int32_t c, const float* gamma,
                         const float* beta, float eps, float momentum,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean,
                         float* invstd, float* scale, float* shift, void* ws,
                         int64_t ws_bytes, usc_stream_t s);
/* y = [relu]( x*scale[c] + shift[c] (+ residual) ); mask-free (backward uses y>0). */
int usc_bn_apply(const float* x, const float* scale, const float* shift,
                 const float* residual, int32_t relu, float* y, int64_t n,
                 int32_t c, usc_stream_t s);
/* BatchNorm backward given saved mean/invstd and channel sums of (dy) and (dy*xhat):
 *   g = relu_mask ? (y_out>0 ? dy : 0) : dy      (applied on the fly when y_out!=NULL)
 *   dx = gamma*invstd*( g - mean_g - xhat*mean_gxhat ),  xhat=(x-mean)*invstd
 * Also writes dres = g when dres != NULL (residual branch gradient). */
int usc_bn_backward_dx(const float* x, const float* dy, const float* y_out,
                       const float* mean, const float* invstd,
                       const float* gamma, const float* mean_g,
                       const float* mean_gxhat, float* dx, float* dres,
                       int64_t n, int32_t c, usc_stream_t s);
/* Reductions needed by the above in one pass: dbeta[c] = sum g, dgamma[c] = sum g*xhat,
 * mean_g = dbeta/n, mean_gxhat = dgamma/n (both 0 when training==0: eval-mode BN
 * treats the statistics as constants).  g applies the ReLU mask of y_out when given.  accumulate=1 adds dgamma / dbeta into the given
 * buffers (the parameters' gradient tensors) instead of overwriting them. */
/* TILE FORM of the training-mode batch norm around a split-K convolution, for maps of up to usc_bn_tile_max_rows() rows
 * (the U-Net's coarse levels, where every launch sits on its latency floor): conv -> BN (+ residual) (+ ReLU) of
 * models/modules/resnet_block.py:48-64 / models/modules/common.py:22 in TWO launches after the convolution instead of
 * four (slice reduction, statistics, finalisation, apply):
 *   forward   y = sum of the G slices `partial` f32[G][n][c] in slice order (G == 0: y already holds the conv output),
 *             per-tile f64 column sums, then every workgroup of the apply launch adds the <= 64 tile sums in tile order,
 *             finalises (mean / invstd / scale / shift and the running statistics written once) and applies
 *             out = [relu](y * scale + shift [+ residual]).  Same statistics as usc_bn_forward_stats up to the order of
 *             the f64 additions.
 *   backward  dout' = (accumulate ? dout : 0) + sum of the G input-gradient slices (G == 0: dout is finished),
 *             g = ReLU-masked dout' (y_out != NULL), tile sums of g and g * xhat; then dgamma / dbeta (added into when
 *             dbn_accumulate), and dy = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)) (training == 0: the two
 *             means are 0).  g is written to dres when given (the residual branch's gradient), else in place into dout
 *             when it was formed from slices; a finished dout without dres is not written.
 * ws: usc_bn_tile_ws_bytes(c) bytes.  usc_bn_tile_ok(n, c): do the kernels cover this map (c a multiple of 32, <= 1024)?
 * usc_bn_tile_max_rows(): the map size up to which usc_conv_bn_act_forward / _backward and usc_program_run TAKE this form
 * (default 4 096 rows; USC3D_BN_TILE_ROWS, 0 = never). */
int64_t usc_bn_tile_max_rows(void);
int usc_bn_tile_ok(int64_t n, int32_t c);
int64_t usc_bn_tile_ws_bytes(int32_t c);
int usc_bn_tile_forward(const float* partial, int32_t G, float* y, int64_t n,
                        int32_t c, const float* gamma, const float* beta,
                        float eps, float momentum, float* running_mean,
                        float* running_var, int64_t* num_batches_tracked,
                        float* mean, float* invstd, float* scale, float* shift,
                        const float* residual, int32_t relu, float* out,
                        void* ws, int64_t ws_bytes, usc_stream_t s);
int usc_bn_tile_backward(const float* partial, int32_t G, int32_t accumulate,
                         float* dout, const float* y, const float* y_out,
                         const float* mean, const float* invstd,
                         const float* gamma, int64_t n, int32_t c,
                         int32_t training, int32_t dbn_accumulate,
                         float* dgamma, float* dbeta, float* dy, float* dres,
                         void* ws, int64_t ws_bytes, usc_stream_t s);
int usc_bn_backward_reduce(const float* x, const float* dy, const float* y_out,
                           const float* mean, const float* invstd, int64_t n,
                           int32_t c, int32_t training, int32_t accumulate, float* dgamma,
                           float* dbeta, float* mean_g, float* mean_gxhat,
                           void* ws, int64_t ws_bytes, usc_stream_t s);
/* y = max(x,0);  dx = (y>0) ? dy : 0 */
int usc_relu_fwd(const float* x, float* y, int64_t numel, usc_stream_t s);
int usc_relu_bwd(const float* y, const float* dy, float* dx, int64_t numel,
                 usc_stream_t s);

/* ------------------------------------------------------------------------
 * P  average pooling — replaces [ME] MinkowskiAvgPooling(kernel_size=2,
 * stride=2, dimension=3) forward (models/mask3d.py:131,213,432).
 *   out[p,:] = mean over present children nbr2[k*n_coarse+p]
 * ---------------------------------------------------------------------- */
int usc_avgpool_down2(const float* in, int32_t c, const int32_t* nbr2,
                      int64_t n_coarse, float* out, usc_stream_t s);
/* The attention-mask chain of the mask module (models/mask3d.py:418-439: per-voxel
 * logits = segment logits gathered through point2segment, pooled
 * num_pooling_steps times, then sigmoid < 0.5) without the [voxels, Q] table:
 *   row_of (optional, i64[n_fine]): child ch reads row row_of[ch] of `in`
 *          (leading dimension ld >= c) — first step, `in` = the [segments, Q] logits;
 *   mask_out (optional, u8[n_coarse, c]): last step writes
 *          sigmoid(mean) < 0.5 instead of the means (`out` may then be NULL). */
int usc_avgpool_down2_ex(const float* in, int32_t c, int32_t ld,
                         const int64_t* row_of, const int32_t* nbr2,
                         int64_t n_coarse, float* out, uint8_t* mask_out,
                         usc_stream_t s);

/* ------------------------------------------------------------------------
 * gather / scatter rows (D1 mask-module row gather mask3d.py:418-419, feature
 * gathers by unique_map datasets/utils.py:412-417)
 * ---------------------------------------------------------------------- */
/* out[i,:] = src[idx[i],:]   idx i64[n] */
int usc_gather_rows(const float* src, int32_t c, const int64_t* idx, int64_t n,
                    float* out, usc_stream_t s);

/* dst[idx[i],:] += src[i,:]  (dst caller-zeroed) — backward of usc_gather_rows for sampled
 * index sets (float atomics; exact and order-independent when idx has no duplicates). */
int usc_scatter_add_rows(const float* src, int32_t c, const int64_t* idx,
                         int64_t n, float* dst, usc_stream_t s);

/* dst[idx[i],:] = src[i,:] for an index set the CALLER knows to be free of duplicates (a permutation,
 * torch.randperm(n)[:k] — the decoder's sampled keys, models/mask3d.py:325): plain 16-byte stores, no atomics;
 * with a zeroed dst this is the backward of usc_gather_rows at gather bandwidth.  Duplicates would race. */
int usc_scatter_rows_unique(const float* src, int32_t c, const int64_t* idx,
                            int64_t n, float* dst, usc_stream_t s);
/* dst[idx[i],:] += src[i,:] for a duplicate-free index set: plain read-modify-write.  Several sampled subsets of one
 * table (the three decoders' key samples of a level, models/mask3d.py:306-349: autograd sums their three scattered
 * gradients) are accumulated launch after launch into ONE buffer instead of three zero-filled tensors and two adds. */
int usc_scatter_rows_unique_add(const float* src, int32_t c, const int64_t* idx,
                                int64_t n, float* dst, usc_stream_t s);

/* The decoder's key sampling of one pass (reference models/mask3d.py:306-346: three row gathers by the sampled
 * indices, `attn[attn.sum(1) == K] = False`, `attn |= padding`), two launches:
 *   out_feats[b,k,:] = feats[idx[b*K+k],:]   out_pos likewise (pos may be NULL)   f32, c and p multiples of 4
 *   out_mask[b,k,:]  = mask[idx[b*K+k],:] (bool bytes [n,q], q <= 128), then
 *     - a query column that is masked in ALL K gathered rows of its scene is cleared in the scene's real rows,
 *     - rows k >= n_valid[b] (padding; their idx repeats a real row) are fully masked.
 * n_valid: HOST array i32[n_scenes] (n_scenes <= 16).  ws: usc_sample_keys_ws_bytes().
 * Partial calls: feats == NULL with c == 0 gathers the mask rows only (the part of a pass's keys that depends on the
 * queries, :337-346); mask == NULL with q == 0 the feature / positional rows only (query-independent: a caller may
 * issue it ahead of the decoder loop on another stream; one launch, no ws). */
int64_t usc_sample_keys_ws_bytes(int32_t n_scenes, int32_t K, int32_t q);
int usc_sample_keys(const float* feats, int32_t c, const uint8_t* mask, int32_t q, const float* pos, int32_t p,
                    const int64_t* idx, int32_t n_scenes, int32_t K, const int32_t* n_valid, float* out_feats,
                    uint8_t* out_mask, float* out_pos, void* ws, int64_t ws_bytes, usc_stream_t s);

/* ------------------------------------------------------------------------
 * Q3  segment mean — replaces torch_scatter.scatter_mean(src, index, dim=0)
 * (models/mask3d.py:12,223; trainer/trainer.py:449) forward + backward.
 * CSR built once per scene: order i64[n] rows sorted by segment (stable),
 * seg_off i64[S+1].
 * ---------------------------------------------------------------------- */
int64_t usc_segment_csr_ws_bytes(int64_t n, int64_t S);
int usc_segment_csr(const int64_t* seg, int64_t n, int64_t S, int64_t* order,
                    int64_t* seg_off, void* ws, int64_t ws_bytes,
                    usc_stream_t s);
int usc_segment_mean_fwd(const float* src, int32_t c, const int64_t* order,
                         const int64_t* seg_off, int64_t S, float* out,
                         usc_stream_t s);
/* dsrc[i,:] = dout[seg[i],:] / count[seg[i]] */
int usc_segment_mean_bwd(const float* dout, int32_t c, const int64_t* seg,
                         const int64_t* seg_off, int64_t n, float* dsrc,
                         usc_stream_t s);

/* Per-segment mean over the NON-ZERO feature rows of each segment (N1:
 * aggregate_features mode 'mean', pseudo_masks/unscene3d_pseudo_main.py:350-402):
 * out f32[S,d], nonzero_cnt i64[S] (may be NULL).  Uses the segment CSR. */
int usc_segment_mean_nonzero(const float* feats, int32_t d,
                             const int64_t* order, const int64_t* seg_off,
                             int64_t S, float* out, int64_t* nonzero_cnt,
                             usc_stream_t s);
/* Same with the per-channel MAX over the non-zero rows (aggregation_mode 'max',
 * unscene3d_pseudo_main.py:366: valid_segment_feats.max(0)[0]); segments without
 * a non-zero row give 0. */
int usc_segment_max_nonzero(const float* feats, int32_t d,
                            const int64_t* order, const int64_t* seg_off,
                            int64_t S, float* out, int64_t* nonzero_cnt,
                            usc_stream_t s);

/* Masked cross attention of the mask decoder, all heads at once:
 *   o = softmax(q k^T / sqrt(16) + mask) v   per (batch, head), head dim 16, L <= 128 queries,
 * q, o f32[L,B,E], k, v f32[S,B,E] (sequence-first, E = 16*H), mask u8[B,S,L] (non-zero =
 * masked, shared by the heads).  No [heads, L, S] tensor is materialised: the forward is split
 * over keys (partial (o, max, sum) per split + a combine pass) and saves lse f32[B*H,128]; the
 * backward recomputes the probabilities and reduces dq partials in a fixed order.
 * Replaces nn.MultiheadAttention's attention core in CrossAttentionLayer
 * (models/mask3d.py:547-605; memory_mask built at :341-348). */
int64_t usc_attn_ws_bytes(int32_t L, int32_t S, int32_t B, int32_t H);
int usc_attn_fwd(const float* q, const float* k, const float* v,
                 const uint8_t* mask, int32_t L, int32_t S, int32_t B, int32_t H,
                 int32_t E, float* o, float* lse, void* ws, int64_t ws_bytes,
                 usc_stream_t s);
int usc_attn_bwd(const float* q, const float* k, const float* v,
                 const uint8_t* mask, const float* o, const float* lse,
                 const float* dO, int32_t L, int32_t S, int32_t B, int32_t H,
                 int32_t E, float* dq, float* dk, float* dv,
                 int32_t mask_bits_in_ws /* ws = the forward call's workspace, packed mask still at its start */,
                 void* ws, int64_t ws_bytes, usc_stream_t s);

/* Self attention of the decoder queries (S = L <= 128 keys, no mask, head dim 16): q, k, v, o, dO, dq, dk, dv
 * f32[L,B,E] sequence-first, lse f32[B*H,128].  ONE launch each way (forward: one workgroup per (batch, head);
 * backward: per (batch, head) one workgroup per query tile for dq and one per key chunk for dk / dv); no partial
 * sum leaves a workgroup and every sum has a fixed order (bit-reproducible under any load, no atomics).
 * Replaces nn.MultiheadAttention's attention core in SelfAttentionLayer (models/mask3d.py:491-545). */
int usc_self_attn_fwd(const float* q, const float* k, const float* v, int32_t L,
                      int32_t B, int32_t H, int32_t E, float* o, float* lse,
                      usc_stream_t s);
int usc_self_attn_bwd(const float* q, const float* k, const float* v, const float* o,
                      const float* lse, const float* dO, int32_t L, int32_t B,
                      int32_t H, int32_t E, float* dq, float* dk, float* dv,
                      usc_stream_t s);

/* Rectangular linear sum assignment (minimise) of n_prob independent cost matrices f32[n_prob, nr, nc], one wave per
 * problem, entirely on the device: row_ind / col_ind i64[n_prob, min(nr,nc)] (rows ascending, like scipy), status
 * i32[n_prob] (0 = solved; 1 = infeasible, i.e. infinite / NaN costs — the indices are then the identity).
 * The algorithm, its f64 dual updates and its tie-breaking are those of scipy.optimize.linear_sum_assignment
 * (rectangular_lsap: shortest augmenting paths), so equal costs give scipy's assignment.
 * Replaces `linear_sum_assignment(C.cpu())` of HungarianMatcher.memory_efficient_forward (models/matcher.py:150-168):
 * no device->host copy and no host solve in the middle of the training step. */
int usc_lsap_batch(const float* cost, int32_t n_prob, int32_t nr, int32_t nc,
                   int64_t* row_ind, int64_t* col_ind, int32_t* status, usc_stream_t s);

/* The set criterion on the device, per scene and for all L <= 16 prediction levels at once (reference
 * models/matcher.py:98-168 cost matrices; models/criterion.py:22-73, :138-216 losses).  masks[l] / dmasks[l]: the
 * level's mask logits f32[S, ld] (ld >= Q columns, Q <= 128 queries) and their gradient; logits: class logits
 * addressed as logits[l*ls_level + q*ls_q + c], c < C; labels i64[T] (253 = ignore), T <= 32 targets.
 *   usc_criterion_target_bits  tm u8[T,S] -> bits u32[S] (bit t = row s belongs to target t), cnt i32[T] = |tm[t]|
 *   usc_criterion_costs        cost = w_mask*BCE + w_class*(-p[label]) + w_dice*dice  f32[L,Q,T] (the LSAP input), its
 *                              parts cmask / cdice [L,Q,T], nmat [L,Q,T] = sum_s sigmoid(x) tm, ssum [L,Q] =
 *                              sum_s sigmoid(x), logp [L,Q,C] = log softmax (kept for the losses and the backward)
 *   usc_criterion_losses       src/tid i64[L,T] (usc_lsap_batch) -> part f32[L,4] = (sum w*nll, sum w, mask loss, dice
 *                              loss) of this scene, tcls i32[L,Q] = target class per query (noobj = C-1 unmatched)
 *   usc_criterion_table        parts [B,L,4] of the batch's scenes -> table [L,4] = (loss_ce, loss_mask, loss_dice, 0),
 *                              den_tot [L]
 *   usc_criterion_backward     gtable = d total / d table [L,4] -> dmasks (full padded width, zero outside the matched
 *                              columns) and dlogits (same addressing as logits)
 * Fixed summation order everywhere.  Replaces the torch op chains of HungarianMatcher.memory_efficient_forward and
 * SetCriterion.loss_labels / loss_masks. */
int64_t usc_criterion_ws_bytes(int32_t L, int32_t S, int32_t T);
int usc_criterion_target_bits(const uint8_t* tm, int32_t T, int32_t S, uint32_t* bits,
                              int32_t* cnt, usc_stream_t s);
int usc_criterion_costs(const float* const* masks, int32_t L, int32_t ld, int32_t S,
                        int32_t Q, int32_t T, const uint32_t* bits, const int32_t* cnt,
                        const float* logits, int64_t ls_level, int64_t ls_q, int32_t C,
                        const int64_t* labels, float w_mask, float w_class, float w_dice,
                        float* cost, float* cmask, float* cdice, float* nmat, float* ssum,
                        float* logp, void* ws, int64_t ws_bytes, usc_stream_t s);
int usc_criterion_losses(const float* cmask, const float* cdice, const float* logp,
                         const int64_t* src, const int64_t* tid, const int64_t* labels,
                         const float* class_w, int32_t L, int32_t Q, int32_t T, int32_t C,
                         int32_t noobj, int32_t* tcls, float* part, usc_stream_t s);
int usc_criterion_table(const float* parts, int32_t B, int32_t L, float* table,
                        float* den_tot, usc_stream_t s);
int usc_criterion_backward(const float* const* masks, float* const* dmasks, int32_t L,
                           int32_t ld, int32_t S, int32_t Q, int32_t T, const uint32_t* bits,
                           const int32_t* cnt, const int64_t* src, const int64_t* tid,
                           const float* nmat, const float* ssum, const float* logp,
                           const int32_t* tcls, const float* class_w, const float* gtable,
                           const float* den_tot, int32_t C, int64_t ls_level, int64_t ls_q,
                           float* dlogits, usc_stream_t s);

/* Linear layer on a handful of rows (the 100 decoder queries):
 *   y[M,N] = x[M,K] W[N,K]^T + b[N]   (b may be NULL);  N, K multiples of 32.
 * usc_linear_bwd: dx[M,K] = dy W, dW[N,K] = dy^T x, db[N] = column sums of dy; each
 * of dx / dW may be NULL to skip it (db is produced with dW); accumulate=1 adds
 * dW / db into the given buffers (parameter gradients).  One workgroup per 32x32
 * output tile on the f32 matrix cores, one launch per product.
 * Replaces nn.Linear / the nn.MultiheadAttention projections of the mask decoder
 * (models/mask3d.py:491-651 attention layers and FFN, :70-72 mask_embed_head). */
int usc_linear_fwd(const float* x, const float* W, const float* b, int32_t M,
                   int32_t N, int32_t K, float* y, usc_stream_t s);
int usc_linear_bwd(const float* dy, const float* x, const float* W, int32_t M,
                   int32_t N, int32_t K, float* dx, float* dW, float* db,
                   int32_t accumulate, usc_stream_t s);
/* The same two launches with what surrounds the layer in the decoder folded in
 * (every extra pointer optional):
 *   x2      the layer's input is x + x2 (query / key positional encodings,
 *           models/mask3d.py:485,517) — forward and weight gradient;
 *   relu    y = max(y, 0) (FFN activation, :542); y_relu = that output: the
 *           gradients count dy only where it is > 0;
 *   dx_add  [M,K] added to dx (a second gradient path into the same input). */
int usc_linear_fwd_ex(const float* x, const float* x2, const float* W,
                      const float* b, int32_t M, int32_t N, int32_t K,
                      int32_t relu, float* y, usc_stream_t s);
/* ... writing y as an [M_pad, N] table whose rows M..M_pad-1 are zero: the 100 query embeddings of the mask module
 * (models/mask3d.py:425) zero-extended to the 128 rows the following product's kernels want, in the producing launch
 * (was: a fill and a copy per call, and a slice copy on the way back). */
int usc_linear_fwd_pad(const float* x, const float* x2, const float* W, const float* b, int32_t M, int32_t N, int32_t K,
                       int32_t relu, float* y, int32_t M_pad, usc_stream_t s);
/* ... several projections of ONE input in one launch: W [N,K] is a stack of N/split_cols weight matrices (the packed
 * in_proj_weight of nn.MultiheadAttention), y comes back as [N/split_cols][M][split_cols] (each projection its own
 * contiguous matrix) and the positional term x2 enters the output columns < x2_cols only — the q, k, v of a
 * self-attention block (models/mask3d.py:507-517: q = k = tgt + query_pos, v = tgt), three launches before. */
int usc_linear_fwd_split(const float* x, const float* x2, const float* W, const float* b, int32_t M, int32_t N, int32_t K,
                         int32_t x2_cols, int32_t split_cols, float* y, usc_stream_t s);
int usc_linear_bwd_ex(const float* dy, const float* y_relu, const float* x,
                      const float* x2, const float* W, int32_t M, int32_t N,
                      int32_t K, float* dx, const float* dx_add, float* dW,
                      float* db, int32_t accumulate, usc_stream_t s);
/* ... and with a second addend and a second output: dx = dy W + dx_add + dx_add2, dx_b = dy W + dx_add (each optional).
 * One product feeding two gradients — the layer input's, which also collects a residual path (dx_add2), and the
 * positional term's, which does not (models/mask3d.py:485-494: `tgt + pos` into the projection, `tgt` into the
 * residual) — without an extra add launch. */
int usc_linear_bwd_ex2(const float* dy, const float* y_relu, const float* x, const float* x2, const float* W, int32_t M,
                       int32_t N, int32_t K, float* dx, const float* dx_add, const float* dx_add2, float* dx_b, float* dW,
                       float* db, int32_t accumulate, usc_stream_t s);
/* Backward of usc_linear_fwd_split for the self-attention projections (split_cols = E, N = 3E, x2_cols = 2E), one launch:
 * dy3 [3][M][E] = dq | dk | dv;  dx = dq Wq + dk Wk + dv Wv (+ dres: the block's residual path);  dx_b (optional) =
 * dq Wq + dk Wk, the positional term's gradient;  dW [3E][E] and db [3E] written or (accumulate) added to, with the
 * positional term in the q and k rows only. */
int usc_qkv_proj_bwd(const float* dy3, const float* x, const float* pos, const float* W, int32_t M, int32_t E, float* dx,
                     float* dx_b, const float* dres, float* dW, float* db, int32_t accumulate, usc_stream_t s);
/* out[c] (+)= sum over the n rows of x f32[n, c]: the bias gradient of a linear
 * layer over many rows (the 3 200 / 12 800 sampled voxels of a decoder pass,
 * models/mask3d.py:351-352 lin_squeeze and the key / value projections of
 * :547-605).  Fixed summation order (64-row partials, then 16 interleaved ascending chains): the same
 * bits on every launch, also when replayed from a captured graph.
 * ws: usc_col_sum_ws_bytes(n, c). */
int64_t usc_col_sum_ws_bytes(int64_t n, int32_t c);
int usc_col_sum(const float* x, int64_t n, int32_t c, float* out,
                int32_t accumulate, void* ws, int64_t ws_bytes, usc_stream_t s);

/* LayerNorm over the last dimension of x f32[rows, d] (d in 64*{1,2,3,4,6,8}):
 *   y = (x - mean) * rstd * gamma + beta,  rstd = 1/sqrt(var + eps)  (biased var);
 * mean/rstd f32[rows] are saved for the backward, which is ONE launch for
 * rows <= 1024 (dx, dgamma, dbeta; column sums reduced in a fixed wave order;
 * accumulate=1 adds dgamma / dbeta into the given buffers).
 * Replaces nn.LayerNorm of the mask decoder (models/mask3d.py:174 decoder_norm,
 * :515/:572/:627 post-norms of SelfAttentionLayer / CrossAttentionLayer / FFNLayer). */
int usc_layernorm_fwd(const float* x, const float* gamma, const float* beta,
                      int64_t rows, int32_t d, float eps, float* y, float* mean,
                      float* rstd, usc_stream_t s);
/* y = LayerNorm(x + res): the post-norm residual of the decoder layers
 * (models/mask3d.py:523-524, 493-494, 543-544) in the same launch; the sum is
 * written to sum_out f32[rows, d], which the backward takes as its `x`. */
int usc_add_layernorm_fwd(const float* x, const float* res, const float* gamma,
                          const float* beta, int64_t rows, int32_t d, float eps,
                          float* y, float* sum_out, float* mean, float* rstd,
                          usc_stream_t s);
int64_t usc_layernorm_bwd_ws_bytes(int64_t rows, int32_t d);
int usc_layernorm_bwd(const float* dy, const float* x, const float* mean,
                      const float* rstd, const float* gamma, int64_t rows,
                      int32_t d, float* dx, float* dgamma, float* dbeta,
                      int32_t accumulate, void* ws, int64_t ws_bytes,
                      usc_stream_t s);
/* The same with dx = (LayerNorm's input gradient) + dx_add (f32[rows, d] or NULL): the gradient that reaches x AROUND
 * the norm — `queries` feed `decoder_norm` inside mask_module AND the next decoder layer (models/mask3d.py:356-373,
 * :410) — summed in this launch instead of by an autograd add. */
int usc_layernorm_bwd_ex(const float* dy, const float* x, const float* mean,
                         const float* rstd, const float* gamma, int64_t rows,
                         int32_t d, const float* dx_add, float* dx, float* dgamma,
                         float* dbeta, int32_t accumulate, void* ws,
                         int64_t ws_bytes, usc_stream_t s);

/* ------------------------------------------------------------------------
 * Q1  furthest point sampling — replaces pointnet2._ext.furthest_point_sampling
 * (third_party/pointnet2/_ext_src/src/sampling.cpp:67-88,
 *  sampling_gpu.cu:73-232; called from models/mask3d.py:228).
 * xyz f32[b,n,3], tmp f32[b,n] pre-filled with 1e10, idx i32[b,m].
 * Bit-exact with the reference kernel's result incl. its tie-break
 * (winner = tied k with smallest (k mod 512), then smallest k, for n>=512)
 * and its |p|^2 <= 1e-3 skip.
 * ---------------------------------------------------------------------- */
int usc_furthest_point_sampling(const float* xyz, int32_t b, int32_t n,
                                int32_t m, float* tmp, int32_t* idx,
                                usc_stream_t s);

/* ------------------------------------------------------------------------
 * Q2  Fourier positional encoding — replaces
 * PositionEmbeddingCoordsSine.get_fourier_embeddings
 * (models/position_embedding.py:128-157) with normalize=True
 * (shift_scale_points :12-40, dst_range [0,1]):
 *   out[i, :] = [sin(2*pi*xn@B) | cos(2*pi*xn@B)],  xn=(x-lo)/(hi-lo)
 * xyz f32[n,3], lo/hi f32[3], gauss_B f32[3,d/2], out f32[n,d] (row-major).
 * ---------------------------------------------------------------------- */
int usc_fourier_posenc(const float* xyz, int64_t n, const float* lo,
                       const float* hi, const float* gauss_B, int32_t d,
                       float* out, usc_stream_t s);

/* ------------------------------------------------------------------------
 * N2-N4  normalized-cut pseudo masks — replaces the numpy/scipy functions of
 * pseudo_masks/unscene3d_pseudo_main.py:82-153 (normalize_mat,
 * get_affinity_matrix, second_smallest_eigenvector) and cosine_sim
 * (utils/freemask_utils.py:8-18).
 * ---------------------------------------------------------------------- */
/* normed f32[S,d] = F.normalize(F) (cosine_mode=1: divided once more by ||.||+1e-9);
 * sim f32[S,S] = normed normed^T; cosine_mode=1 also applies cosine_sim's per-row
 * min-max (attn -= rowmin; attn /= rowmax + 1e-9). */
int usc_ncut_similarity(const float* F, int64_t S, int32_t d,
                        int32_t cosine_mode, float* normed, float* sim,
                        usc_stream_t s);
/* The same over F with the rows flagged in zero_rows (u8[S] or NULL) multiplied by 0 first: the features
 * get_masked_affinity_matrix (unscene3d_pseudo_main.py:122-135) hands to the next iteration are
 * `(1 - painting) * feats`, painting only grows, so iteration i sees the ORIGINAL features with the rows painted so
 * far zeroed — formed inside the row normalisation instead of by four element-wise passes per iteration. */
int usc_ncut_similarity_masked(const float* F, const uint8_t* zero_rows, int64_t S,
                               int32_t d, int32_t cosine_mode, float* normed,
                               float* sim, usc_stream_t s);
/* In-place normalize_mat: A -= min(A[A != 0]) if any(A > 0); A[A < 0] = 0;
 * A /= max(A) + 1e-5.  ws >= 12288 bytes. */
int usc_ncut_normalize_mat(float* A, int64_t S, void* ws, int64_t ws_bytes,
                           usc_stream_t s);
/* Abin u8[S,S] = ((simA [+ simB]) / (1|2)) > tau with painted rows/columns forced
 * off (unscene3d_pseudo_main.py:426-427); deg f64[S] = column sums of
 * (A ? 1 : eps) BEFORE the painting overwrite (:111-118).  simB, painted may be NULL. */
int usc_ncut_binarize(const float* simA, const float* simB, int64_t S, float tau,
                      double eps, const uint8_t* painted, uint8_t* Abin,
                      double* deg, usc_stream_t s);
/* Eigenvector #2 (second smallest eigenvalue) of (D - A) v = lambda D v, i.e.
 * scipy.linalg.eigh(D - A, D, subset_by_index=[1, 2])[1][:, 0], fp64, with LAPACK's
 * sign (dsygvx sequence restated: lower triangle, dsytrd 'L' reflectors, dstein
 * normalisation).  A = Abin ? 1 : eps.  evec f64[S], eval f64[2] = (lambda_1, lambda_2). */
int64_t usc_ncut_fiedler_ws_bytes(int64_t S);
int usc_ncut_fiedler(const uint8_t* Abin, const double* deg, int64_t S, double eps,
                     double* evec, double* eval, void* ws, int64_t ws_bytes,
                     usc_stream_t s);

/* ------------------------------------------------------------------------
 * N6  exact 1-nearest-neighbour — replaces scipy.spatial.KDTree.query(k=1)
 * (pseudo_masks/unscene3d_pseudo_main.py:341-343, :651-652).  f64 distances on
 * f32 inputs; ties resolve to the lowest reference index.
 * query f32[nq,3], ref f32[nr,3] -> idx i64[nq], dist2 f32[nq] (may be NULL).
 * ---------------------------------------------------------------------- */
int usc_knn1(const float* query, int64_t nq, const float* ref, int64_t nr,
             int64_t* idx, float* dist2, usc_stream_t s);

/* ------------------------------------------------------------------------
 * E1  eps-ball connected components — replaces
 * sklearn.cluster.DBSCAN(eps, min_samples=1).fit(xyz).labels_
 * (trainer/trainer.py:521-523): min-label propagation with pointer jumping; the
 * caller iterates usc_cc_eps_step (ping-pong label buffers) until `changed`
 * (i32[1], device) reads 0, then usc_cc_eps_finish numbers the components in
 * first-seen order (labels i64[n]).  xyz f32[n,3].
 * ---------------------------------------------------------------------- */
int usc_cc_eps_init(int32_t* label, int64_t n, usc_stream_t s);
int usc_cc_eps_step(const float* xyz, int64_t n, float eps,
                    const int32_t* label_in, int32_t* label_out,
                    int32_t* changed, usc_stream_t s);
int usc_cc_eps_finish(const int32_t* label, int64_t n, int32_t* rank_ws,
                      int64_t* labels, usc_stream_t s);

/* ------------------------------------------------------------------------
 * L3  tri-plane projection — replaces
 * custom_cuda_utils.project_sparse_voxels_to_planes{,_backward}
 * (utils/cuda_utils/cuda_utils.cpp:26-46, cuda_utils_kernel.cu:371-603; wrapper
 * models/noise_robust_loss.py:16-71).  In-place accumulation into caller-zeroed
 * outputs, like the reference; voxels outside [0,dim) are dropped like there.
 * coords i32[V,4] (b,x,y,z) shifted to >= 0, pred/target f32[V,inst];
 * planes f32[dim_a,dim_b,inst], counts i32[dim_a,dim_b].
 * Backward: grad_pred[v,p] = mean of the non-zero among the three plane grads.
 * ---------------------------------------------------------------------- */
int usc_project_planes_fwd(const int32_t* coords, const float* pred,
                           const float* target, int64_t V, int32_t inst,
                           int32_t dim_x, int32_t dim_y, int32_t dim_z,
                           float* pred_xy, float* pred_xz, float* pred_yz,
                           float* tgt_xy, float* tgt_xz, float* tgt_yz,
                           int32_t* cnt_xy, int32_t* cnt_xz, int32_t* cnt_yz,
                           usc_stream_t s);
int usc_project_planes_bwd(const int32_t* coords, int64_t V, int32_t inst,
                           int32_t dim_x, int32_t dim_y, int32_t dim_z,
                           const float* g_xy, const float* g_xz,
                           const float* g_yz, float* grad_pred,
                           usc_stream_t s);

/* ------------------------------------------------------------------------
 * F1  2D -> 3D feature projection — replaces [CUDA ext] project_features_cuda.project_features_cuda
 * and project_features_cuda.unproject_depth_images (utils/cuda_utils/project_image_cuda.cpp:10-31,
 * project_image_cuda_kernel.cu:24-146,190-290; wrapper utils/cuda_utils/raycast_image.py:18-77; caller
 * pseudo_masks/unscene3d_pseudo_main.py:287-330).
 * views f32[B,V,4,4] row-major camera-to-grid matrices in voxel units (already shifted by the per-batch
 * minimum like the wrapper does), intrinsics f32[B,4] = fx, fy, mx, my.
 * Every pixel (b,v,y,x) marches  t = t0, t0+inc, ...  (sequential fp32 accumulation, like the reference)
 * and reports the row of the first occupied voxel at round-half-away(cam + t*dir):
 *   hit i32[B,V,H,W]   row, -1 = none.  Row 0 never hits (the reference's grid stores 0 for "empty").
 *   seg i64[B,V,H,W]   optional: row, or n_rows for a miss — the segment ids usc_segment_csr takes.
 * _map: the occupancy is the coordinate hash of usc_coordmap_build (tensor stride 1); shift i32[B,3] is
 *       the per-batch minimum coordinate (grid voxel + shift = map coordinate).
 * _dense: occupancy i64[B,dim_z,dim_y,dim_x] exactly as the reference builds it (0 = empty).
 * ---------------------------------------------------------------------- */
/* Optional free-space filter of the _map ray cast: one bit per 8x8x8-voxel brick of the (shifted) grid,
 * set where a voxel row > 0 lives; mask u32[B, ceil(bricks_x*bricks_y*bricks_z / 32)] (zeroed here),
 * brick index = (gz*bricks_y + gy)*bricks_x + gx.  coords i32[n,4] are the map's coordinates. */
int usc_brick_mask_build(const int32_t* coords, int64_t n, const int32_t* shift, int32_t B,
                         int32_t bricks_x, int32_t bricks_y, int32_t bricks_z,
                         uint32_t* mask, usc_stream_t s);
int usc_raycast_first_hit_map(const uint64_t* table_keys, const int32_t* table_vals,
                              int64_t cap, int64_t n_rows, const int32_t* shift,
                              const uint32_t* brick_mask /* may be NULL */, int32_t bricks_x,
                              int32_t bricks_y, int32_t bricks_z,
                              const float* views, const float* intrinsics,
                              int32_t B, int32_t V, int32_t H, int32_t W,
                              float depth_min, float depth_max, float ray_increment,
                              int32_t* hit, int64_t* seg, usc_stream_t s);
int usc_raycast_first_hit_dense(const int64_t* occupancy, int32_t dim_z, int32_t dim_y,
                                int32_t dim_x, int64_t n_rows, const float* views,
                                const float* intrinsics, int32_t B, int32_t V,
                                int32_t H, int32_t W, float depth_min, float depth_max,
                                float ray_increment, int32_t* hit, int64_t* seg,
                                usc_stream_t s);
/* Per-voxel reduction of the hit pixels' features over the CSR (order, seg_off) that
 * usc_segment_csr built from `seg` with n_rows+1 segments; feats f32[n_pix,c]; sums run in
 * ascending pixel order (deterministic; the reference's float atomics are unordered).
 *   mode 0: out f32[n_rows,c] = sum / (count + 10e-5) on every row (0 where nothing hit);
 *           num i32[n_rows] = count                                  (raycast_image.py:66-68)
 *   mode 1: out is the scene feature table: out[r] = (out[r] + sum/(count+10e-5)) / 2 on the rows
 *           that were hit only; num = count     (unscene3d_pseudo_main.py:311-313, :327-328)
 *   mode 2: out[r] += sum, num[r] += count — the operator's own in-place accumulation
 *           (project_image_cuda_kernel.cu:49,61-64)
 *   num may be NULL. */
int usc_project_reduce(const float* feats, int32_t c, const int64_t* order,
                       const int64_t* seg_off, int64_t n_rows, int32_t mode,
                       float* out, int32_t* num, usc_stream_t s);
/* Prediction mode (project_image_cuda_kernel.cu:68-109): out i32[n_rows,c] (caller-initialised to the
 * ignore label) = max(out, preds of the pixels that hit the row); preds i32[n_pix,c]. */
int usc_project_predictions(const int32_t* preds, int32_t c, const int32_t* hit,
                            int64_t n_pix, int32_t* out, usc_stream_t s);
/* depth f32[V,H,W], views f32[V,4,4], intrinsics f32[V,4] -> cloud f32[V*H*W,5] = (view, pixel index,
 * x, y, z) for depth > 0; other rows are left as the caller initialised them
 * (project_image_cuda_kernel.cu:249-290). */
int usc_unproject_depth(const float* depth, const float* views, const float* intrinsics,
                        int32_t V, int32_t H, int32_t W, float* cloud, usc_stream_t s);

/* ------------------------------------------------------------------------
 * T  optimizer step — torch.optim.AdamW(lr, betas, eps, weight_decay) of
 * `configure_optimizers` (trainer/trainer.py:953-966) over FLAT f32 buffers of n elements
 * (16-byte aligned): p *= 1 - lr*wd; m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2;
 * p -= (lr / (1 - b1^step)) * m / (sqrt(v) / sqrt(1 - b2^step) + eps).  step counts from 1.
 * ---------------------------------------------------------------------- */
/* Diagnostic, no reference counterpart: `wgs` workgroups that spin for `microseconds` on stream s.  HIP multiplexes its
 * streams onto a few hardware queues (4 per priority by default), and two streams that share one do not overlap however
 * independent their work is: unscene3d_amd/streams.py times pairs of these launches to pick streams that really run
 * beside the compute stream. */
int usc_spin(int64_t microseconds, int wgs, usc_stream_t s);

int usc_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                   int64_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int64_t step, usc_stream_t s);

/* ------------------------------------------------------------------------
 * A1  elastic distortion of the training augmentation — the per-point half of
 * datasets/semseg.py:651-688 `elastic_distortion` (called from freemask_semseg.py:356-361):
 * xyz_out[i,0:3] = xyz_in[i,0:3] + magnitude * trilinear(noise)(xyz_in[i,0:3]), evaluated in f64 like
 * scipy's RegularGridInterpolator(bounds_error=0, fill_value=0).  xyz f32 or f64 [n,row_stride] (only the
 * first three columns are read/written; in place allowed), noise f32[dim_x,dim_y,dim_z,3] (already
 * smoothed), axis_* f64[dim_*] the grid coordinates (np.linspace of the reference).
 * ---------------------------------------------------------------------- */
int usc_elastic_displace(const void* xyz_in, int32_t is_f64, int64_t n, int32_t row_stride,
                         const float* noise, int32_t dim_x, int32_t dim_y, int32_t dim_z,
                         const double* axis_x, const double* axis_y, const double* axis_z,
                         double magnitude, void* xyz_out, usc_stream_t s);

/* ------------------------------------------------------------------------
 * F4  training-time augmentations of the scene reader — replaces the numpy /
 * volumentations / albumentations steps of datasets/freemask_semseg.py:334-406.
 * ---------------------------------------------------------------------- */
/* x[:, 0:3] <- x[:, 0:3] M^T + t in place (f64 arithmetic, one rounding to the
 * table's type): centring + random shift (:335-342), axis flips (:348-351),
 * Scale3d and RotateAroundAxis3d of conf/augmentation/volumentations_aug.yaml.
 * M (9, row-major) and t (3) are HOST doubles; x is f32 or f64 [n, row_stride]. */
int usc_affine_rows(void* x, int32_t is_f64, int64_t n, int32_t row_stride,
                    const double* M, const double* t, usc_stream_t s);
/* out[c] = ((x[0,c] + x[1,c]) + x[2,c]) + ... in f32, c < cols <= 64: numpy's
 * summation order for a.sum(0) / a.mean(0) of a C-contiguous table, so that
 * `coordinates -= coordinates.mean(0)` (:335) is reproduced bit for bit. */
int usc_colsum_sequential(const float* x, int64_t n, int32_t row_stride,
                          int32_t cols, float* out, usc_stream_t s);
/* out[i,c] = lut[c*256 + uint8(color[i,c])], c < 3: RandomBrightnessContrast /
 * RGBShift of conf/augmentation/albumentations_aug.yaml and the colour
 * normalisation (:408-409) as per-channel tables over the uint8-truncated
 * colours.  color f32[n,row_stride], lut f32[3][256] (device), out f32. */
int usc_color_lut(const float* color, int64_t n, int32_t row_stride,
                  const float* lut, float* out, int32_t out_stride,
                  usc_stream_t s);

/* ------------------------------------------------------------------------
 * F2  Felzenszwalb mesh over-segmentation — replaces the reference's
 * felzenszwalb_cpp extension (utils/cpp_utils/segmentator.cpp:17-154
 * segment_graph / segment_mesh; caller pseudo_masks/datasets/scannet.py:156-197).
 * Device: face normals, vertex normals (the running blend of face normals in
 * face order, :62-82), edge weights (:85-121) — bit-equal to the reference
 * (separately rounded operations).  The caller sorts the 3F edges by weight on
 * the device (stable) and hands the sorted lists to usc_felz_merge_host for the
 * two sequential merge loops (:17-44, :127-139).
 * ---------------------------------------------------------------------- */
/* face_normals f32[F,3] = normalised cross(p2-p1, p3-p1). */
int usc_felz_face_normals(const float* vertices, const int32_t* faces,
                          int64_t n_faces, float* face_normals, usc_stream_t s);
/* corner_order i64[3F]: the (face, corner) entries 3f+j sorted by their vertex,
 * stable (usc_segment_csr over faces.reshape(-1)); vertex_off i64[V+1]. */
int usc_felz_vertex_normals(const float* face_normals, const int64_t* corner_order,
                            const int64_t* vertex_off, int64_t n_vertices,
                            float* normals, usc_stream_t s);
/* Edge 3f+0 = (i1,i2), 3f+1 = (i1,i3), 3f+2 = (i3,i2): endpoints and weight. */
int usc_felz_edge_weights(const float* vertices, const float* colors,
                          const float* normals, const int32_t* faces,
                          int64_t n_faces, int32_t* edge_a, int32_t* edge_b,
                          float* weights, usc_stream_t s);
/* HOST pointers (the one exception in this ABI): edges sorted by weight ->
 * comps i32[V] = union-find representative of every vertex. */
int usc_felz_merge_host(const int32_t* edge_a, const int32_t* edge_b,
                        const float* weights, int64_t n_edges, int32_t n_vertices,
                        float kthr, int32_t seg_min_verts, int32_t* comps);

#ifdef __cplusplus
}
#endif
#endif /* USC3D_H */
