// units.hip — native issue path for the backbone: one C call launches every kernel of a
// "convolution -> batch norm (+ residual) (+ ReLU)" unit, forward or backward.
//
// The kernels are the ones behind the per-operator entry points (usc_spconv_*, usc_bn_*); what moves here
// is the ISSUE LOOP.  From Python every launch cost ~10-20 us of interpreter, ctypes, allocator and autograd
// bookkeeping, ~1 000 of them per training step for the backbone alone, which put the host level with the
// device.  A unit is 3-4 launches forward and 4-6 backward behind ONE call; the kernel choice
// (mask-sorted / tile-compacted / row-order / pair-list form, weight transposes, split counts) that lived in
// unscene3d_amd/ops.py is restated here so that both paths launch identical kernels.
//
// Reference: models/modules/resnet_block.py:48-64 (conv -> norm -> relu / `out += residual`),
// models/res16unet.py:231-297 (conv{0..4} / convtr{4..7} + bn + relu), models/modules/common.py:125-188.
#include <stdlib.h>

#include <vector>

#include "common.h"

using namespace usc;

namespace {

struct WsCursor {
  char* base; int64_t bytes; int64_t off;
  void* take(int64_t n) {
    n = align_up(n > 0 ? n : 16, 256);
    if (off + n > bytes) return nullptr;
    void* p = base + off;
    off += n;
    return p;
  }
  int64_t left() const { return bytes - off; }
  void* rest() { return base + off; }
};

inline bool table_is_compact(int64_t n_out, int cin, int cout, int K) {
  return (usc_spconv_plan(0, n_out, cin, cout, K) >> 12) & 1;
}
inline bool sorted_ok(const usc_kmap* m, int cin, int cout) {
  return m->nbr && m->perm && m->tile_mask && m->K > 1 && m->K <= 32 && cin % 32 == 0 && cout % 32 == 0 && cin <= 4096;
}

// bytes the table-form GEMM needs (out rows n_out, W possibly to be transposed first)
int64_t table_gemm_ws(const usc_kmap* m, int64_t n_out, int cin, int cout, int K, bool want_transposed) {
  if (m->nbr && sorted_ok(m, cin, cout) && !table_is_compact(n_out, cin, cout, K))
    return align_up(usc_spconv_sorted_ws_bytes(n_out, cin, cout, K), 256);
  int64_t b = align_up(usc_spconv_gather_gemm_ws_bytes(n_out, cin, cout, K), 256);
  if (want_transposed && !(m->nbr && table_is_compact(n_out, cin, cout, K))) b += align_up((int64_t)K * cin * cout * 4, 256);
  return b;
}

// out[o] (+)= sum_k in[nbr[k][o]] W[k]; `wt`: W is given as the forward [K, cout, cin] and the mirrored transpose
// W'[k][c][n] = W[K-1-k][n][c] is meant (mirror only when K > 1).  `nbr` NULL: identity rows (1x1).
// slices the split-K launch left un-reduced for whoever consumes `out` next (usc_spconv_sorted_gemm_ex)
struct Slices { const float* partial = nullptr; int G = 0; };

int table_gemm(const usc_kmap* m, const int32_t* nbr, const float* in, int64_t n_in, int cin, const float* W, int K,
               int cout, int64_t n_out, const float* bias, float* out, int accumulate, int wt, WsCursor ws,
               usc_stream_t s, Slices* left = nullptr) {
  if (left) *left = Slices{};
  if (n_out == 0) return USC_OK;
  const bool compact = nbr && table_is_compact(n_out, cin, cout, K);
  if (nbr && nbr == m->nbr && sorted_ok(m, cin, cout) && !compact) {
    const int64_t b = usc_spconv_sorted_ws_bytes(n_out, cin, cout, K);
    void* w = b > 0 ? ws.take(b) : nullptr;
    USC_REQUIRE(b == 0 || w, "usc unit: workspace too small (sorted gemm)");
    int32_t G = 0;
    const int rc = usc_spconv_sorted_gemm_ex(in, n_in, cin, W, K, cout, nbr, m->perm, m->tile_mask, n_out, bias, out,
                                             accumulate, wt, w, b, left ? &G : nullptr, s);
    if (!rc && left && G > 0) { left->partial = (const float*)w; left->G = G; }
    return rc;
  }
  if (wt && !compact) {
    float* Wt = (float*)ws.take((int64_t)K * cin * cout * 4);
    USC_REQUIRE(Wt, "usc unit: workspace too small (weight transpose)");
    // W is [K, cout, cin] as a forward weight; the product wants [K, cin, cout]
    int rc = usc_weight_transpose(W, K, cout, cin, K > 1 ? 1 : 0, Wt, s);
    if (rc) return rc;
    W = Wt;
    wt = 0;
  }
  const int64_t b = usc_spconv_gather_gemm_ws_bytes(n_out, cin, cout, K);
  void* w = b > 0 ? ws.take(b) : nullptr;
  USC_REQUIRE(b == 0 || w, "usc unit: workspace too small (gather gemm)");
  return usc_spconv_gather_gemm(in, n_in, cin, W, K, cout, nbr, n_out, bias, out, accumulate, wt, w, b, s);
}

// ---- fork/join of the weight gradient onto a side stream --------------------------------------------------------
// The input-gradient and the weight-gradient kernels of one convolution are independent.  On the coarse levels of
// the U-Net (hundreds to a few thousand rows) both are latency-bound launches that occupy a fraction of the 256 CUs,
// so running them side by side costs about as much as the longer of the two.  The side stream is the CALLER's
// (usc_set_side_stream); the fork and the join are events inside the one call, so nothing outlives it: when the
// call returns, everything it queued is ordered before whatever the caller queues next on its stream.
struct SideStream { hipStream_t st = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
SideStream g_side[16];
constexpr int64_t kForkMaxRows = 24576;   // larger maps fill the chip on their own (and the tile-compacted kernel
                                          // sizes its tiles for whole rounds of 256 CUs)

// ---- the weight-gradient lane -----------------------------------------------------------------------------------
// The join of the fork above put the caller's stream behind the side stream once per convolution, and that cost more
// than the overlap returned.  The lane never joins inside a call: the weight gradient of a small map is queued on the
// lane stream behind ONE event of the caller's stream (everything it reads is ready at that point) and the call
// returns; the input-gradient chain — the critical path of the backward pass — carries on, and the latency-bound
// weight-gradient launches of the coarse levels fill the CUs it leaves idle.  The caller joins once, when the
// gradients are needed (usc_wgrad_lane_join), keeps x / dy / dW alive until then, and gives the lane its own scratch.
// One weight gradient that has been handed to the lane but not launched yet (usc_wgrad_lane_hold).
struct HeldWgrad { const float* x; const float* dy; float* dW; const int32_t* a_idx; const int32_t* b_idx; const int64_t* koff;
                   int64_t rows, ws_bytes; int cin, cout, K; };
struct Lane { hipStream_t st = nullptr; hipEvent_t fork = nullptr, join = nullptr; void* ws = nullptr; int64_t ws_bytes = 0;
              int64_t max_rows = 0; bool dirty = false;
              // hold: weight gradients of maps with >= hold_min_rows rows are only NOTED until the caller's chain reaches a map
              // with <= release_max_rows rows (or the hold is lifted); from then on everything goes to the lane at once
              bool hold = false; int64_t hold_min_rows = 0, release_max_rows = 0; std::vector<HeldWgrad> held; };
Lane g_lane[16];
Lane* lane_for_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  return g_lane[dev].st ? &g_lane[dev] : nullptr;
}

// the held weight gradients go to the lane, in the order they were noted, behind one event of the caller's stream
int lane_release(Lane* lane, usc_stream_t s) {
  lane->hold = false;
  if (lane->held.empty()) return USC_OK;
  if (hipEventRecord(lane->fork, as_stream(s)) != hipSuccess || hipStreamWaitEvent(lane->st, lane->fork, 0) != hipSuccess) {
    lane->held.clear();
    set_error("usc wgrad lane: releasing the held weight gradients failed (event)");
    return USC_ERR_LAUNCH;
  }
  int rc = USC_OK;
  lane->dirty = true;
  for (const HeldWgrad& h : lane->held) {
    rc = usc_spconv_wgrad(h.x, h.cin, h.dy, h.cout, h.K, h.a_idx, h.b_idx, h.koff, h.rows, h.dW, 1, lane->ws, h.ws_bytes,
                          (usc_stream_t)lane->st);
    if (rc) break;
  }
  lane->held.clear();
  return rc;
}

SideStream* side_for_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  return g_side[dev].st ? &g_side[dev] : nullptr;
}

struct ConvShape { int64_t n_in, n_out; };   // rows of the convolution's input / output feature matrices
inline ConvShape conv_shape(const usc_kmap* m, int kind) {
  return kind == USC_CONV_UP ? ConvShape{m->n_out, m->n_in} : ConvShape{m->n_in, m->n_out};
}

int check_map(const usc_kmap* m, int kind, int cin, int cout, const char* who) {
  USC_REQUIRE(m, "%s: null kernel map", who);
  USC_REQUIRE(kind == USC_CONV_SAME || kind == USC_CONV_DOWN || kind == USC_CONV_UP, "%s: unknown conv kind %d", who, kind);
  USC_REQUIRE(cin >= 1 && cout >= 1 && m->K >= 1 && m->n_in >= 0 && m->n_out >= 0, "%s: bad sizes", who);
  USC_REQUIRE(m->nbr || (m->K == 1 && kind == USC_CONV_SAME), "%s: K>1 needs a neighbour table", who);
  USC_REQUIRE(kind != USC_CONV_SAME || m->n_in == m->n_out, "%s: a stride-1 map has n_in == n_out", who);
  return USC_OK;
}


// ---- element-wise steps of a step program ---------------------------------------------------------------------------
// dst[r][0:ca] = a[r][:], dst[r][ca:ca+cb] = b[r][:]   (float4 lanes; ca, cb multiples of 4)
__global__ __launch_bounds__(256) void cat_cols_kernel(const float4* __restrict__ a, int ca4, const float4* __restrict__ b,
                                                      int cb4, float4* __restrict__ dst, int64_t total4) {
  const int c4 = ca4 + cb4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / c4;
    const int j = (int)(e - r * c4);
    dst[e] = j < ca4 ? a[r * ca4 + j] : b[r * cb4 + (j - ca4)];
  }
}
// da[r][:] = src[r][0:ca], db[r][:] (+)= src[r][ca:ca+cb]
__global__ __launch_bounds__(256) void split_cols_kernel(const float4* __restrict__ src, int ca4, int cb4, float4* __restrict__ da,
                                                        float4* __restrict__ db, int accumulate, int64_t total4) {
  const int c4 = ca4 + cb4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / c4;
    const int j = (int)(e - r * c4);
    const float4 v = src[e];
    if (j < ca4) {
      da[r * ca4 + j] = v;
    } else {
      float4* o = db + r * cb4 + (j - ca4);
      if (accumulate) { const float4 t = *o; *o = make_float4(t.x + v.x, t.y + v.y, t.z + v.z, t.w + v.w); }
      else *o = v;
    }
  }
}
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
    float4 d = reinterpret_cast<float4*>(dst)[e];
    const float4 v = reinterpret_cast<const float4*>(src)[e];
    d.x += v.x; d.y += v.y; d.z += v.z; d.w += v.w;
    reinterpret_cast<float4*>(dst)[e] = d;
  }
  for (int64_t e = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) dst[e] += src[e];
}
inline unsigned ew_grid(int64_t work) {
  int64_t g = ceil_div(work, (int64_t)256 * 4);
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (unsigned)g;
}

// weight gradients queued during ONE usc_program_run call (same kernel map and shape -> one grouped launch)
struct DeferredWgrads {
  const usc_kmap* map = nullptr; int cin = 0, cout = 0, n = 0;
  const float* a[16]; const float* b[16]; float* dW[16];
};
int flush_deferred(DeferredWgrads& q, void* ws, int64_t ws_bytes, usc_stream_t s) {
  if (q.n == 0) return USC_OK;
  const usc_kmap* m = q.map;
  int rc = USC_OK;
  if (q.n >= 2 && usc_spconv_wgrad_group_ok(q.n, q.cin, q.cout, m->K, m->pair_capacity)) {
    rc = usc_spconv_wgrad_group(q.n, q.a, q.b, q.dW, q.cin, q.cout, m->K, m->pair_in, m->pair_out, m->koff, m->pair_capacity, 1, s);
  } else {
    const int64_t b = usc_spconv_wgrad_ws_bytes_rows(m->K, q.cin, q.cout, m->pair_capacity);
    if (b > ws_bytes) { set_error("usc_program_run: workspace too small (weight gradient)"); rc = USC_ERR_ARG; }
    for (int r = 0; r < q.n && !rc; ++r)
      rc = usc_spconv_wgrad(q.a[r], q.cin, q.b[r], q.cout, m->K, m->pair_in, m->pair_out, m->koff, m->pair_capacity, q.dW[r], 1, ws, b, s);
  }
  q.n = 0;
  q.map = nullptr;
  return rc;
}
}  // namespace

extern "C" {

int usc_set_side_stream(usc_stream_t side) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
    set_error("usc_set_side_stream: unsupported device index");
    return USC_ERR_ARG;
  }
  SideStream& e = g_side[dev];
  if (!side) {
    e.st = nullptr;
    return USC_OK;
  }
  if (!e.fork) {
    if (hipEventCreateWithFlags(&e.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e.join, hipEventDisableTiming) != hipSuccess) {
      set_error("usc_set_side_stream: hipEventCreate failed");
      return USC_ERR_LAUNCH;
    }
  }
  e.st = as_stream(side);
  return USC_OK;
}

int usc_set_wgrad_lane(usc_stream_t lane, void* lane_ws, int64_t lane_ws_bytes, int64_t max_rows) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
    set_error("usc_set_wgrad_lane: unsupported device index");
    return USC_ERR_ARG;
  }
  Lane& e = g_lane[dev];
  if (!lane) {
    e.st = nullptr;
    e.held.clear();
    e.hold = false;
    return USC_OK;
  }
  USC_REQUIRE(lane_ws && lane_ws_bytes > 0 && max_rows > 0, "usc_set_wgrad_lane: the lane needs its own scratch and a row bound");
  if (!e.fork) {
    if (hipEventCreateWithFlags(&e.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e.join, hipEventDisableTiming) != hipSuccess) {
      set_error("usc_set_wgrad_lane: hipEventCreate failed");
      return USC_ERR_LAUNCH;
    }
  }
  e.st = as_stream(lane);
  e.ws = lane_ws;
  e.ws_bytes = lane_ws_bytes;
  e.max_rows = max_rows;
  e.dirty = false;
  return USC_OK;
}

int usc_wgrad_lane_hold(int32_t mode, int64_t hold_min_rows, int64_t release_max_rows, usc_stream_t s) {
  Lane* lane = lane_for_current_device();
  if (!lane) return USC_OK;
  if (mode > 0) {
    USC_REQUIRE(hold_min_rows > release_max_rows && release_max_rows >= 0, "usc_wgrad_lane_hold: hold_min_rows must exceed release_max_rows");
    if (!lane->held.empty()) { const int rc = lane_release(lane, s); if (rc) return rc; }
    lane->hold = true;
    lane->hold_min_rows = hold_min_rows;
    lane->release_max_rows = release_max_rows;
    return USC_OK;
  }
  if (mode < 0) {            // the caller's pass failed: what was noted points into buffers that are gone
    lane->held.clear();
    lane->hold = false;
    return USC_OK;
  }
  return lane_release(lane, s);
}

int32_t usc_wgrad_lane_holding(void) {
  Lane* lane = lane_for_current_device();
  return lane && lane->hold ? 1 : 0;
}

int usc_wgrad_lane_join(usc_stream_t s) {
  Lane* lane = lane_for_current_device();
  if (lane && !lane->held.empty()) { const int rc = lane_release(lane, s); if (rc) return rc; }
  if (!lane || !lane->dirty) return USC_OK;
  if (hipEventRecord(lane->join, lane->st) != hipSuccess || hipStreamWaitEvent(as_stream(s), lane->join, 0) != hipSuccess) {
    set_error("usc_wgrad_lane_join: joining the lane failed");
    return USC_ERR_LAUNCH;
  }
  lane->dirty = false;
  return USC_OK;
}

static void conv_ws_parts(const usc_kmap* m, int kind, int cin, int cout, int64_t* fwd, int64_t* dgrad, int64_t* wg) {
  const int K = m->K;
  const ConvShape sh = conv_shape(m, kind);
  const int64_t wbytes = align_up((int64_t)K * cin * cout * 4, 256);
  if (kind == USC_CONV_SAME) {
    *fwd = table_gemm_ws(m, sh.n_out, cin, cout, K, false);
    *dgrad = table_gemm_ws(m, sh.n_in, cout, cin, K, true);
  } else if (kind == USC_CONV_DOWN) {
    *fwd = table_gemm_ws(m, sh.n_out, cin, cout, K, false);
    *dgrad = wbytes;                                            // transposed weights for the pair-list form
  } else {
    *fwd = 0;                                                   // pair-list form, no scratch
    *dgrad = wbytes + table_gemm_ws(m, sh.n_in, cout, cin, K, false);
  }
  *dgrad = align_up(*dgrad + 256, 256);
  const int64_t rows = m->nbr ? m->pair_capacity : sh.n_in;
  *wg = align_up(usc_spconv_wgrad_ws_bytes_rows(K, cin, cout, rows), 256);
  if (kind == USC_CONV_SAME && m->nbr && cin <= 4 && cout == 32) {      // the stem: table-form weight gradient
    const int64_t t = align_up(usc_spconv_wgrad_table_ws_bytes(K, cin, cout), 256);
    if (t > *wg) *wg = t;
  }
}

int64_t usc_conv_ws_bytes(const usc_kmap* m, int32_t kind, int32_t cin, int32_t cout) {
  if (!m) return 0;
  int64_t fwd, dgrad, wg;
  conv_ws_parts(m, kind, cin, cout, &fwd, &dgrad, &wg);
  const int64_t bwd = dgrad + wg;          // disjoint regions: the two may run on different streams
  return (fwd > bwd ? fwd : bwd) + 256;
}

static int conv_forward_impl(const usc_kmap* m, int32_t kind, const float* x, int32_t cin, const float* W, int32_t cout,
                             const float* bias, float* y, void* ws, int64_t ws_bytes, usc_stream_t s, Slices* left);

int usc_conv_forward(const usc_kmap* m, int32_t kind, const float* x, int32_t cin, const float* W, int32_t cout,
                     const float* bias, float* y, void* ws, int64_t ws_bytes, usc_stream_t s) {
  return conv_forward_impl(m, kind, x, cin, W, cout, bias, y, ws, ws_bytes, s, nullptr);
}

static int conv_forward_impl(const usc_kmap* m, int32_t kind, const float* x, int32_t cin, const float* W, int32_t cout,
                             const float* bias, float* y, void* ws, int64_t ws_bytes, usc_stream_t s, Slices* left) {
  if (left) *left = Slices{};
  int rc = check_map(m, kind, cin, cout, "usc_conv_forward");
  if (rc) return rc;
  const ConvShape sh = conv_shape(m, kind);
  if (sh.n_out == 0) return USC_OK;
  USC_REQUIRE(x && W && y, "usc_conv_forward: null pointer");
  WsCursor cur{(char*)ws, ws ? ws_bytes : 0, 0};
  if (kind == USC_CONV_UP) {
    // every fine row has exactly one parent: out[fine] = in[coarse] W[k]
    USC_REQUIRE(m->pair_in && m->pair_out && m->koff, "usc_conv_forward: transposed conv needs the pair lists");
    USC_REQUIRE(!bias, "usc_conv_forward: bias is not supported on the pair-list form");
    return usc_spconv_pairs_gemm(x, cin, W, m->K, cout, m->pair_out, m->pair_in, m->koff, sh.n_out, y, s);
  }
  return table_gemm(m, m->nbr, x, sh.n_in, cin, W, m->K, cout, sh.n_out, bias, y, 0, 0, cur, s, left);
}

// the input gradient of a convolution whose split-K slices are still to be summed: dx = (accumulate ? dx : 0) + sum of
// the G slices; consumed by the batch norm backward of the unit whose output gradient dx is (tile form), or reduced by
// flush_pending
struct Pending { float* dx = nullptr; const float* partial = nullptr; int G = 0; int64_t n = 0; int c = 0; int accumulate = 0; };

static int flush_pending(Pending& p, usc_stream_t s) {
  if (p.G <= 0) return USC_OK;
  const int rc = usc_group_reduce(p.partial, p.G, p.n, p.c, nullptr, p.accumulate, p.dx, s);
  p = Pending{};
  return rc;
}

static bool tile_form_on() {
  static const bool on = usc_bn_tile_max_rows() > 0;
  return on;
}
// the unit calls take the tile form on maps of up to usc_bn_tile_max_rows() rows (the kernels themselves cover any size)
static bool tile_rows_ok(int64_t n, int c) { return tile_form_on() && n <= usc_bn_tile_max_rows() && usc_bn_tile_ok(n, c); }

static int conv_backward_impl(const usc_kmap* m, int32_t kind, const float* x, int32_t cin, const float* W, int32_t cout,
                              const float* dy, float* dx, int32_t dx_accumulate, float* dW, int32_t dW_accumulate, void* ws,
                              int64_t ws_bytes, usc_stream_t s, Pending* out);

int usc_conv_backward(const usc_kmap* m, int32_t kind, const float* x, int32_t cin, const float* W, int32_t cout,
                      const float* dy, float* dx, int32_t dx_accumulate, float* dW, int32_t dW_accumulate, void* ws,
                      int64_t ws_bytes, usc_stream_t s) {
  return conv_backward_impl(m, kind, x, cin, W, cout, dy, dx, dx_accumulate, dW, dW_accumulate, ws, ws_bytes, s, nullptr);
}

static int conv_backward_impl(const usc_kmap* m, int32_t kind, const float* x, int32_t cin, const float* W, int32_t cout,
                              const float* dy, float* dx, int32_t dx_accumulate, float* dW, int32_t dW_accumulate, void* ws,
                              int64_t ws_bytes, usc_stream_t s, Pending* out) {
  if (out) *out = Pending{};
  int rc = check_map(m, kind, cin, cout, "usc_conv_backward");
  if (rc) return rc;
  const ConvShape sh = conv_shape(m, kind);
  const int K = m->K;
  USC_REQUIRE(x && W && dy, "usc_conv_backward: null pointer");
  WsCursor cur{(char*)ws, ws ? ws_bytes : 0, 0};
  // lane: the weight gradient queued on the lane stream, joined by the caller at the end of the backward pass
  if (dW && dW_accumulate && x && dy) {
    Lane* lane = lane_for_current_device();
    const int64_t rows = m->nbr ? m->pair_capacity : sh.n_in;
    const int64_t b = usc_spconv_wgrad_ws_bytes_rows(K, cin, cout, rows);
    // USC3D_WGRAD_LANE_MIN_ROWS (experiment knob): only maps with at least that many rows go to the lane — the
    // throughput-bound weight gradients of the fine levels, queued to run beside the latency-bound coarse-level chain
    static const int64_t lane_min_rows = getenv("USC3D_WGRAD_LANE_MIN_ROWS") ? atoll(getenv("USC3D_WGRAD_LANE_MIN_ROWS")) : 0;
    const bool lane_ok = lane && sh.n_in > 0 && sh.n_in >= lane_min_rows && sh.n_in <= lane->max_rows && sh.n_out <= lane->max_rows &&
                         b <= lane->ws_bytes && (!m->nbr || (m->pair_in && m->pair_out && m->koff));
    const int64_t big = sh.n_in > sh.n_out ? sh.n_in : sh.n_out;
    // the chain has reached a map too small to fill the chip: what was held runs beside it from here on
    if (lane && lane->hold && big <= lane->release_max_rows) { rc = lane_release(lane, s); if (rc) return rc; }
    if (lane_ok && lane->hold && big >= lane->hold_min_rows) {
      HeldWgrad h{x, dy, dW, nullptr, nullptr, nullptr, m->nbr ? rows : sh.n_in, b, cin, cout, m->nbr ? K : 1};
      if (m->nbr) {
        h.a_idx = kind == USC_CONV_UP ? m->pair_out : m->pair_in;
        h.b_idx = kind == USC_CONV_UP ? m->pair_in : m->pair_out;
        h.koff = m->koff;
      }
      lane->held.push_back(h);
      dW = nullptr;          // noted; launched by lane_release
    } else if (lane_ok &&
        hipEventRecord(lane->fork, as_stream(s)) == hipSuccess && hipStreamWaitEvent(lane->st, lane->fork, 0) == hipSuccess) {
      usc_stream_t ls = (usc_stream_t)lane->st;
      lane->dirty = true;
      // the stem (3 -> 32 channels) keeps its table-form kernel on the lane too: it is the LAST weight gradient of the
      // backward pass — the lane's tail, which the step waits for — and the pair-list kernel needs 110 us for it, the table
      // form 57 (round 6: the lane branch used to send everything through usc_spconv_wgrad)
      static const bool stem_table = !getenv("USC3D_STEM_KERNEL") || atoi(getenv("USC3D_STEM_KERNEL")) != 0;
      const int64_t bt = usc_spconv_wgrad_table_ws_bytes(K, cin, cout);
      if (stem_table && kind == USC_CONV_SAME && m->nbr && cin <= 4 && cout == 32 && K <= 32 && bt <= lane->ws_bytes)
        rc = usc_spconv_wgrad_table(x, cin, dy, cout, m->nbr, K, sh.n_out, dW, 1, lane->ws, bt, ls);
      else if (!m->nbr)
        rc = usc_spconv_wgrad(x, cin, dy, cout, 1, nullptr, nullptr, nullptr, sh.n_in, dW, 1, lane->ws, b, ls);
      else if (kind == USC_CONV_UP)
        rc = usc_spconv_wgrad(x, cin, dy, cout, K, m->pair_out, m->pair_in, m->koff, rows, dW, 1, lane->ws, b, ls);
      else
        rc = usc_spconv_wgrad(x, cin, dy, cout, K, m->pair_in, m->pair_out, m->koff, rows, dW, 1, lane->ws, b, ls);
      if (rc) return rc;
      dW = nullptr;          // done (queued); the rest of this call is the input gradient on the caller's stream
    }
  }
  // fork: the weight gradient on the side stream while this stream runs the input gradient
  SideStream* side = (dx && dW && sh.n_in > 0 && sh.n_in <= kForkMaxRows && sh.n_out <= kForkMaxRows) ? side_for_current_device() : nullptr;
  hipStream_t wst = as_stream(s);
  int64_t dgrad_bytes = 0;
  // slices may stay behind only when the consumer can take them (tile-form batch norm on the input map) and nothing in
  // this call overwrites them: the weight gradient then takes its scratch behind the input gradient's region
  Slices left;
  const bool may_defer = out && dx && !side && tile_rows_ok(sh.n_in, cin);
  if (may_defer) {
    int64_t f_, w_;
    conv_ws_parts(m, kind, cin, cout, &f_, &dgrad_bytes, &w_);
    if (dgrad_bytes + w_ > cur.bytes) dgrad_bytes = 0;
  }
  Slices* want = (may_defer && dgrad_bytes > 0) ? &left : nullptr;
  if (side) {
    // disjoint scratch: [0, dgrad_bytes) for this stream, the rest for the side stream
    int64_t f_, w_;
    conv_ws_parts(m, kind, cin, cout, &f_, &dgrad_bytes, &w_);
    if (dgrad_bytes + w_ > cur.bytes || hipEventRecord(side->fork, as_stream(s)) != hipSuccess ||
        hipStreamWaitEvent(side->st, side->fork, 0) != hipSuccess) {
      side = nullptr;
      dgrad_bytes = 0;
    } else {
      wst = side->st;
    }
  }
  if (dx && sh.n_in > 0) {
    if (kind == USC_CONV_SAME) {
      // stride-1 map: the mirrored offset reaches the rows that read row i; transpose folded where the kernel can
      rc = table_gemm(m, m->nbr, dy, sh.n_out, cout, W, K, cin, sh.n_in, nullptr, dx, dx_accumulate, 1, cur, s, want);
    } else if (kind == USC_CONV_DOWN) {
      USC_REQUIRE(!dx_accumulate, "usc_conv_backward: accumulate is not supported on the pair-list form");
      USC_REQUIRE(m->pair_in && m->pair_out && m->koff, "usc_conv_backward: strided conv needs the pair lists");
      float* Wt = (float*)cur.take((int64_t)K * cin * cout * 4);
      USC_REQUIRE(Wt, "usc_conv_backward: workspace too small");
      rc = usc_weight_transpose(W, K, cin, cout, 0, Wt, s);
      if (!rc) rc = usc_spconv_pairs_gemm(dy, cout, Wt, K, cin, m->pair_out, m->pair_in, m->koff, sh.n_in, dx, s);
    } else {
      float* Wt = (float*)cur.take((int64_t)K * cin * cout * 4);
      USC_REQUIRE(Wt, "usc_conv_backward: workspace too small");
      rc = usc_weight_transpose(W, K, cin, cout, 0, Wt, s);
      // dx[coarse] = sum_k dy[child k of coarse] W[k]^T: the child table again, gather form
      if (!rc) rc = table_gemm(m, m->nbr, dy, sh.n_out, cout, Wt, K, cin, sh.n_in, nullptr, dx, dx_accumulate, 0, cur, s, want);
    }
    if (!rc && left.G > 0) {
      out->dx = dx; out->partial = left.partial; out->G = left.G; out->n = sh.n_in; out->c = cin; out->accumulate = dx_accumulate;
    }
    if (rc) dW = nullptr;   // skip the weight gradient, still join below
  }
  if (dW) {
    // same stream: the weight gradient reuses the scratch from the start (stream order); forked: its own region
    WsCursor wc{(char*)ws, ws ? ws_bytes : 0, (side || left.G > 0) ? dgrad_bytes : 0};
    const int64_t rows = m->nbr ? m->pair_capacity : sh.n_in;
    const int64_t b = usc_spconv_wgrad_ws_bytes_rows(K, cin, cout, rows);
    void* w = wc.take(b);
    if (!w) set_error("usc_conv_backward: workspace too small (weight gradient)");
    usc_stream_t ws_stream = (usc_stream_t)wst;
    static const bool stem_table = !getenv("USC3D_STEM_KERNEL") || atoi(getenv("USC3D_STEM_KERNEL")) != 0;
    const int64_t bt = usc_spconv_wgrad_table_ws_bytes(K, cin, cout);
    if (!w) rc = USC_ERR_ARG;
    else if (stem_table && kind == USC_CONV_SAME && m->nbr && cin <= 4 && cout == 32 && K <= 32 && (char*)w + bt <= (char*)ws + ws_bytes)
      rc = usc_spconv_wgrad_table(x, cin, dy, cout, m->nbr, K, sh.n_out, dW, dW_accumulate, w, bt, ws_stream);
    else if (!m->nbr)
      rc = usc_spconv_wgrad(x, cin, dy, cout, 1, nullptr, nullptr, nullptr, sh.n_in, dW, dW_accumulate, w, b, ws_stream);
    else if (kind == USC_CONV_UP)
      rc = usc_spconv_wgrad(x, cin, dy, cout, K, m->pair_out, m->pair_in, m->koff, rows, dW, dW_accumulate, w, b, ws_stream);
    else
      rc = usc_spconv_wgrad(x, cin, dy, cout, K, m->pair_in, m->pair_out, m->koff, rows, dW, dW_accumulate, w, b, ws_stream);
  }
  if (side) {   // join (also on an error above: the caller's stream must not run ahead of the side stream)
    if (hipEventRecord(side->join, side->st) != hipSuccess || hipStreamWaitEvent(as_stream(s), side->join, 0) != hipSuccess) {
      set_error("usc_conv_backward: joining the side stream failed");
      return USC_ERR_LAUNCH;
    }
  }
  return rc;
}

int64_t usc_unit_ws_bytes(const usc_kmap* m, int32_t kind, int32_t cin, int32_t cout) {
  return usc_conv_ws_bytes(m, kind, cin, cout) + align_up(usc_colstats_ws_bytes(0, cout), 256) +
         align_up(2 * (int64_t)cout * 4, 256) + 256;
}

int usc_conv_bn_act_forward(const usc_kmap* m, int32_t kind, const float* x, int32_t cin, const float* W, int32_t cout,
                            const usc_bn* bn, const float* residual, int32_t relu, float* y, float* stats, float* out,
                            void* ws, int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(bn && bn->c == cout, "usc_conv_bn_act_forward: batch-norm width must equal the conv's output width");
  USC_REQUIRE(y && stats && out && bn->gamma && bn->beta, "usc_conv_bn_act_forward: null pointer");
  const ConvShape sh = conv_shape(m, kind);
  WsCursor cur{(char*)ws, ws ? ws_bytes : 0, 0};
  const int64_t sb = usc_colstats_ws_bytes(sh.n_out, cout);
  void* sws = cur.take(sb);                       // BN partials first: the conv's scratch takes the rest
  USC_REQUIRE(sws, "usc_conv_bn_act_forward: workspace too small");
  // tile form (coarse levels): the convolution leaves its split-K slices behind, two launches do the rest
  const bool tile = bn->training && tile_rows_ok(sh.n_out, cout) && sb >= usc_bn_tile_ws_bytes(cout);
  Slices left;
  int rc = conv_forward_impl(m, kind, x, cin, W, cout, nullptr, y, cur.rest(), cur.left(), s, tile ? &left : nullptr);
  if (rc) return rc;
  if (sh.n_out == 0) return USC_OK;
  float* mean = stats, *invstd = stats + cout, *scale = stats + 2 * cout, *shift = stats + 3 * cout;
  if (tile)
    return usc_bn_tile_forward(left.partial, left.G, y, sh.n_out, cout, bn->gamma, bn->beta, bn->eps, bn->momentum,
                               bn->running_mean, bn->running_var, bn->num_batches_tracked, mean, invstd, scale, shift,
                               residual, relu, out, sws, sb, s);
  if (bn->training) {
    rc = usc_bn_forward_stats(y, sh.n_out, cout, bn->gamma, bn->beta, bn->eps, bn->momentum, bn->running_mean,
                              bn->running_var, bn->num_batches_tracked, mean, invstd, scale, shift, sws, sb, s);
  } else {
    USC_REQUIRE(bn->running_mean && bn->running_var, "usc_conv_bn_act_forward: eval mode needs running statistics");
    rc = usc_bn_eval_stats(bn->gamma, bn->beta, bn->running_mean, bn->running_var, bn->eps, cout, mean, invstd, scale,
                           shift, s);
  }
  if (rc) return rc;
  return usc_bn_apply(y, scale, shift, residual, relu, out, sh.n_out, cout, s);
}

static int unit_backward_impl(const usc_kmap* m, int32_t kind, const float* x, int32_t cin, const float* W, int32_t cout,
                              const usc_bn* bn, const float* y, const float* stats, const float* out_relu,
                              const float* dout, float* dy, float* dres, float* dx, int32_t dx_accumulate, float* dW,
                              int32_t dW_accumulate, float* dgamma, float* dbeta, int32_t dbn_accumulate, void* ws,
                              int64_t ws_bytes, usc_stream_t s, Pending* in, Pending* out);

int usc_conv_bn_act_backward(const usc_kmap* m, int32_t kind, const float* x, int32_t cin, const float* W, int32_t cout,
                             const usc_bn* bn, const float* y, const float* stats, const float* out_relu,
                             const float* dout, float* dy, float* dres, float* dx, int32_t dx_accumulate, float* dW,
                             int32_t dW_accumulate, float* dgamma, float* dbeta, int32_t dbn_accumulate, void* ws,
                             int64_t ws_bytes, usc_stream_t s) {
  return unit_backward_impl(m, kind, x, cin, W, cout, bn, y, stats, out_relu, dout, dy, dres, dx, dx_accumulate, dW,
                            dW_accumulate, dgamma, dbeta, dbn_accumulate, ws, ws_bytes, s, nullptr, nullptr);
}

// in: slices of THIS unit's output gradient (in->dx == dout) left by the previous step, or NULL / empty;
// out: receives the slices of this unit's input gradient when they may stay un-reduced (else left empty)
static int unit_backward_impl(const usc_kmap* m, int32_t kind, const float* x, int32_t cin, const float* W, int32_t cout,
                              const usc_bn* bn, const float* y, const float* stats, const float* out_relu,
                              const float* dout, float* dy, float* dres, float* dx, int32_t dx_accumulate, float* dW,
                              int32_t dW_accumulate, float* dgamma, float* dbeta, int32_t dbn_accumulate, void* ws,
                              int64_t ws_bytes, usc_stream_t s, Pending* in, Pending* out) {
  if (out) *out = Pending{};
  USC_REQUIRE(bn && bn->c == cout, "usc_conv_bn_act_backward: batch-norm width must equal the conv's output width");
  USC_REQUIRE(y && stats && dout && dy && dgamma && dbeta && bn->gamma, "usc_conv_bn_act_backward: null pointer");
  const ConvShape sh = conv_shape(m, kind);
  if (sh.n_out == 0) return USC_OK;
  WsCursor cur{(char*)ws, ws ? ws_bytes : 0, 0};
  const int64_t sb = usc_colstats_ws_bytes(sh.n_out, cout);
  void* sws = cur.take(sb);
  float* red = (float*)cur.take(2 * (int64_t)cout * 4);   // mean_g | mean_gxhat
  USC_REQUIRE(sws && red, "usc_conv_bn_act_backward: workspace too small");
  const float* mean = stats, *invstd = stats + cout;
  int rc;
  const bool has_in = in && in->G > 0;
  if (has_in) USC_REQUIRE(in->dx == dout && in->n == sh.n_out && in->c == cout, "usc unit: pending slices do not belong to this unit");
  if (tile_rows_ok(sh.n_out, cout) && sb >= usc_bn_tile_ws_bytes(cout)) {
    rc = usc_bn_tile_backward(has_in ? in->partial : nullptr, has_in ? in->G : 0, has_in ? in->accumulate : 0, (float*)dout, y,
                              out_relu, mean, invstd, bn->gamma, sh.n_out, cout, bn->training, dbn_accumulate, dgamma,
                              dbeta, dy, dres, sws, sb, s);
    if (has_in) *in = Pending{};
    if (rc) return rc;
  } else {
    if (has_in) { rc = flush_pending(*in, s); if (rc) return rc; }
    rc = usc_bn_backward_reduce(y, dout, out_relu, mean, invstd, sh.n_out, cout, bn->training, dbn_accumulate, dgamma,
                                dbeta, red, red + cout, sws, sb, s);
    if (rc) return rc;
    rc = usc_bn_backward_dx(y, dout, out_relu, mean, invstd, bn->gamma, red, red + cout, dy, dres, sh.n_out, cout, s);
    if (rc) return rc;
  }
  return conv_backward_impl(m, kind, x, cin, W, cout, dy, dx, dx_accumulate, dW, dW_accumulate, cur.rest(), cur.left(), s, out);
}

int32_t usc_step_size(void) { return (int32_t)sizeof(usc_step); }

static int64_t program_half_bytes(const usc_step* steps, int32_t n_steps) {
  int64_t need = 256;
  if (!steps) return need;
  for (int i = 0; i < n_steps; ++i) {
    const usc_step& t = steps[i];
    if ((t.op == USC_STEP_UNIT_FWD || t.op == USC_STEP_UNIT_BWD) && t.map) {
      const int64_t b = usc_unit_ws_bytes(t.map, t.kind, t.cin, t.cout);
      if (b > need) need = b;
    }
  }
  return align_up(need, 256);
}

// two halves: step i works in half i & 1, so the input-gradient slices a backward step leaves for the next one survive it
int64_t usc_program_ws_bytes(const usc_step* steps, int32_t n_steps) { return 2 * program_half_bytes(steps, n_steps) + 256; }

int usc_program_run(const usc_step* steps, int32_t begin, int32_t end, void* ws_all, int64_t ws_all_bytes, usc_stream_t s) {
  USC_REQUIRE(steps && begin >= 0 && end >= begin, "usc_program_run: bad step range");
  hipStream_t st = as_stream(s);
  DeferredWgrads q;
  Pending pend;
  int rc = USC_OK;
  // the scratch in two halves (see usc_program_ws_bytes); a caller that sized it for one step only gets the old behaviour
  const int64_t half = ws_all_bytes > 256 ? (((ws_all_bytes - 256) / 2) & ~(int64_t)255) : 0;
  const bool two = ws_all && half >= program_half_bytes(steps + begin, end - begin);
  void* ws = ws_all;
  int64_t ws_bytes = two ? half : ws_all_bytes;
  for (int i = begin; i < end && !rc; ++i) {
    const usc_step& t = steps[i];
    if (two) ws = (char*)ws_all + (i & 1) * half;
    // slices left by the previous step: only the backward of the unit whose output gradient they are can take them
    if (pend.G > 0 && !(t.op == USC_STEP_UNIT_BWD && t.dout == pend.dx)) {
      rc = flush_pending(pend, s);
      if (rc) break;
    }
    switch (t.op) {
      case USC_STEP_UNIT_FWD:
        rc = usc_conv_bn_act_forward(t.map, t.kind, t.x, t.cin, t.W, t.cout, t.bn, t.residual, t.relu, t.y, t.stats, t.out,
                                     ws, ws_bytes, s);
        break;
      case USC_STEP_UNIT_BWD: {
        USC_REQUIRE(t.map, "usc_program_run: step %d has no kernel map", i);
        // the same rule as units.py::_defer_wgrad: only shapes the grouped launch can take are postponed (the 3-channel
        // stem keeps its table-form kernel, fine-level convolutions that can never group run at once)
        const bool defer = t.defer_wgrad && t.dW && t.dW_accumulate && t.kind == USC_CONV_SAME && t.map->K > 1 && t.map->pair_in &&
                           t.cin % 32 == 0 && t.cout % 32 == 0 &&
                           (usc_spconv_wgrad_group_ok(2, t.cin, t.cout, t.map->K, t.map->pair_capacity) ||
                            usc_spconv_wgrad_group_ok(usc_spconv_wgrad_group_max(), t.cin, t.cout, t.map->K, t.map->pair_capacity));
        if (defer && q.n > 0 && (q.map != t.map || q.cin != t.cin || q.cout != t.cout || q.n == 16)) {
          rc = flush_deferred(q, ws, ws_bytes, s);       // (this step's half: the previous step's slices are in the other)
          if (rc) break;
        }
        Pending next;
        rc = unit_backward_impl(t.map, t.kind, t.x, t.cin, t.W, t.cout, t.bn, t.y, t.stats, t.out, t.dout, t.dy, t.dres,
                                t.dx, t.dx_accumulate, defer ? nullptr : t.dW, t.dW_accumulate, t.dgamma, t.dbeta,
                                t.dbn_accumulate, ws, ws_bytes, s, &pend, two ? &next : nullptr);
        if (!rc && pend.G > 0) rc = flush_pending(pend, s);       // (not consumed: cannot happen, kept for safety)
        pend = next;
        if (!rc && defer) {          // x and this step's own dy stay valid until the end of the call (caller's arenas)
          q.map = t.map; q.cin = t.cin; q.cout = t.cout;
          q.a[q.n] = t.x; q.b[q.n] = t.dy; q.dW[q.n] = t.dW;
          ++q.n;
        }
        break;
      }
      case USC_STEP_CAT: {
        USC_REQUIRE(t.a && t.b && t.dst && t.ca % 4 == 0 && t.cb % 4 == 0 && t.n >= 0, "usc_program_run: bad cat step %d", i);
        const int64_t total4 = t.n * ((t.ca + t.cb) / 4);
        if (total4 > 0)
          hipLaunchKernelGGL(cat_cols_kernel, dim3(ew_grid(total4)), dim3(256), 0, st, (const float4*)t.a, t.ca / 4,
                             (const float4*)t.b, t.cb / 4, (float4*)t.dst, total4);
        break;
      }
      case USC_STEP_SPLIT: {
        USC_REQUIRE(t.a && t.dst && t.dst2 && t.ca % 4 == 0 && t.cb % 4 == 0 && t.n >= 0, "usc_program_run: bad split step %d", i);
        const int64_t total4 = t.n * ((t.ca + t.cb) / 4);
        if (total4 > 0)
          hipLaunchKernelGGL(split_cols_kernel, dim3(ew_grid(total4)), dim3(256), 0, st, (const float4*)t.a, t.ca / 4, t.cb / 4,
                             (float4*)t.dst, (float4*)t.dst2, (int)t.accumulate, total4);
        break;
      }
      case USC_STEP_ADD:
        USC_REQUIRE(t.a && t.dst && t.n >= 0, "usc_program_run: bad add step %d", i);
        if (t.n > 0) hipLaunchKernelGGL(add_inplace_kernel, dim3(ew_grid(t.n / 4 + 1)), dim3(256), 0, st, t.dst, t.a, t.n);
        break;
      default:
        set_error("usc_program_run: unknown step op");
        rc = USC_ERR_ARG;
    }
  }
  const int rc3 = flush_pending(pend, s);
  // the queued weight gradients last: with two halves, the one the pending slices (just reduced, in stream order) were not in
  const int rc2 = flush_deferred(q, ws, ws_bytes, s);
  if (rc) return rc;
  if (rc3) return rc3;
  if (rc2) return rc2;
  USC_CHECK_LAUNCH("usc_program_run");
  return USC_OK;
}

}  // extern "C"
