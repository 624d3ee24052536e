// criterion.hip — the set criterion of the self-training step on the device (reference models/matcher.py:98-168,
// models/criterion.py:22-73, :138-216): the Hungarian assignment of every prediction level without leaving the GPU.
//
// usc_lsap_batch: rectangular linear sum assignment (minimise) of a batch of independent [nr, nc] float32 cost
// matrices, one wave per problem.  The reference calls scipy.optimize.linear_sum_assignment on the host
// (matcher.py:161-163: one device->host copy and one single-threaded solve per level and scene).  This is the same
// algorithm — scipy's rectangular_lsap (Crouse's shortest-augmenting-path variant of Jonker-Volgenant): the wider
// side becomes the columns, rows are augmented one at a time by a Dijkstra-like scan over the remaining columns, dual
// variables u / v in f64 — restated so that the RESULT is scipy's also when costs tie:
//   * the scan runs over the `remaining` list in scipy's order (initialised nc-1 ... 0, removal = swap with the last);
//   * scipy's sequential rule "strictly lower wins; an equal value replaces the candidate only if its column is
//     unassigned" means: among the entries equal to the minimum, the LAST unassigned one in list order if there is
//     one, else the first entry.  Each lane keeps (min, first position, last unassigned position) over its strided
//     share of the list; three wave reductions give the same pick as the sequential scan;
//   * f64 arithmetic in scipy's operation order ((minVal + c) - u) - v, no contraction (adds only).
// 100 queries x <= 25 targets: <= 25 augmentations of a few scan steps, ~60 cycles each with the matrix staged in
// LDS: a few microseconds for the 13 levels of a scene side by side, and no host round trip in the middle of the step.
#include "common.h"

#include <limits.h>
#include <stdlib.h>

namespace usc {
namespace {

__device__ inline double wave_min_f64(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmin(x, __shfl_xor(x, o, 64));
  return x;
}
__device__ inline int wave_min_i32(int x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(x, o, 64); x = y < x ? y : x; }
  return x;
}
__device__ inline int wave_max_i32(int x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(x, o, 64); x = y > x ? y : x; }
  return x;
}

struct LsapLayout {
  int64_t u, v, spc, path, row4col, remaining, col4row, SR, SC, cm, total;
};
__host__ __device__ inline LsapLayout lsap_layout(int nr, int nc, bool stage) {
  LsapLayout L;
  int64_t o = 0;
  L.u = o; o += 8 * (int64_t)nr;
  L.v = o; o += 8 * (int64_t)nc;
  L.spc = o; o += 8 * (int64_t)nc;
  L.path = o; o += 4 * (int64_t)nc;
  L.row4col = o; o += 4 * (int64_t)nc;
  L.remaining = o; o += 4 * (int64_t)nc;
  L.col4row = o; o += 4 * (int64_t)nr;
  L.SR = o; o += (nr + 3) / 4 * 4;
  L.SC = o; o += (nc + 3) / 4 * 4;
  L.cm = o; o += stage ? 4 * (int64_t)nr * nc : 0;
  L.total = o;
  return L;
}
constexpr int64_t kLsapLdsBudget = 60 * 1024;

// ---- wave reductions on the DPP path (row shifts + row broadcasts: VALU latency; __shfl_xor is two LDS-crossbar trips
// per level, and a scan step is a chain of three dependent reductions).  A lane without a source keeps its own value
// (bound_ctrl off, old = own), which is the identity of min / max.
template <int CTRL, int ROW_MASK>
__device__ inline int dpp_keep_i32(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xf, false); }
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_keep_f64(double x) {
  const int lo = dpp_keep_i32<CTRL, ROW_MASK>(__double2loint(x)), hi = dpp_keep_i32<CTRL, ROW_MASK>(__double2hiint(x));
  return __hiloint2double(hi, lo);
}
#define USC_DPP_REDUCE(T, OP, DPP, BCAST)                                         \
  x = OP(x, DPP<0x111, 0xf>(x)); x = OP(x, DPP<0x112, 0xf>(x));                   \
  x = OP(x, DPP<0x114, 0xf>(x)); x = OP(x, DPP<0x118, 0xf>(x));                   \
  x = OP(x, DPP<0x142, 0xa>(x)); x = OP(x, DPP<0x143, 0xc>(x));
__device__ inline int imin(int a, int b) { return a < b ? a : b; }
__device__ inline int imax(int a, int b) { return a > b ? a : b; }
__device__ inline double wave_min_f64_dpp(double x) {
  USC_DPP_REDUCE(double, fmin, dpp_keep_f64, 0)
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}
__device__ inline int wave_min_i32_dpp(int x) {
  USC_DPP_REDUCE(int, imin, dpp_keep_i32, 0)
  return __builtin_amdgcn_readlane(x, 63);
}
__device__ inline int wave_max_i32_dpp(int x) {
  USC_DPP_REDUCE(int, imax, dpp_keep_i32, 0)
  return __builtin_amdgcn_readlane(x, 63);
}
#undef USC_DPP_REDUCE

// Register-resident variant for nc <= 128 columns (the decoder's 100 queries): lane l owns the columns l and l + 64 —
// dual variable v, shortest-path cost, assigned row, predecessor and the column's POSITION in scipy's `remaining` list
// all live in registers; only u, col4row, the visited-row flags and the staged cost matrix are in LDS.  Removing a
// column (`remaining[index] = remaining[--num]`) = the lane owning the column at the last position takes over `index`.
// A scan step is then two LDS reads, a handful of f64 operations and three DPP reductions (~400 cycles; the
// LDS-resident form below needs ~1500).
__global__ __launch_bounds__(64) void lsap_reg_kernel(const float* __restrict__ cost_all, int nr0, int nc0,
                                                      int64_t* __restrict__ row_ind, int64_t* __restrict__ col_ind,
                                                      int32_t* __restrict__ status) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int lane = threadIdx.x;
  const bool tr = nr0 > nc0;
  const int nr = tr ? nc0 : nr0, nc = tr ? nr0 : nc0;
  const float* __restrict__ cost = cost_all + (int64_t)blockIdx.x * nr0 * nc0;
  const LsapLayout L = lsap_layout(nr, nc, true);
  double* u = (double*)(lds + L.u);
  double* spc_l = (double*)(lds + L.spc);       // shortest-path costs of the finished augmentation (for the u update)
  int* col4row = (int*)(lds + L.col4row);
  unsigned char* SR = lds + L.SR;
  float* cm = (float*)(lds + L.cm);
  int64_t* rout = row_ind + (int64_t)blockIdx.x * nr;
  int64_t* cout_ = col_ind + (int64_t)blockIdx.x * nr;
  constexpr int NCL = 2;
  double v[NCL], spc[NCL];
  int r4c[NCL], path[NCL], pos[NCL];
  bool sc[NCL];
#pragma unroll
  for (int c = 0; c < NCL; ++c) { v[c] = 0.0; spc[c] = INFINITY; r4c[c] = -1; path[c] = -1; pos[c] = -1; sc[c] = false; }
  for (int e = lane; e < nr; e += 64) { u[e] = 0.0; col4row[e] = -1; }
  // scipy rejects a matrix with a NaN or -inf entry up front ("matrix contains invalid numeric entries"), wherever it
  // sits; the search below would simply never pick such an entry
  bool invalid = false;
  for (int e = lane; e < nr * nc; e += 64) {
    const int i = e / nc, j = e - i * nc;
    const float x = tr ? cost[(int64_t)j * nc0 + i] : cost[(int64_t)i * nc0 + j];
    invalid |= (x != x) || x == -INFINITY;
    cm[e] = x;
  }
  __syncthreads();
  bool infeasible = __ballot(invalid) != 0ull;
  for (int cur = 0; cur < nr && !infeasible; ++cur) {
    double minVal = 0.0;
    int num = nc;
#pragma unroll
    for (int c = 0; c < NCL; ++c) {
      const int j = lane + 64 * c;
      pos[c] = j < nc ? nc - 1 - j : -1;              // remaining[it] = nc - it - 1  <=>  pos(j) = nc - 1 - j
      spc[c] = INFINITY;
      sc[c] = false;
    }
    for (int e = lane; e < nr; e += 64) SR[e] = 0;
    __syncthreads();
    int i = cur, sink = -1;
    while (sink < 0) {
      if (lane == 0) SR[i] = 1;
      const double ui = u[i];
      const float* crow = cm + i * nc;
      double lmin = INFINITY;
      int lfirst = INT_MAX, lastU = -1;
#pragma unroll
      for (int c = 0; c < NCL; ++c) {
        if (pos[c] >= 0) {
          const double r = ((minVal + (double)crow[lane + 64 * c]) - ui) - v[c];
          if (r < spc[c]) { path[c] = i; spc[c] = r; }
          const double sv = spc[c];
          const bool un = r4c[c] < 0;
          if (sv < lmin) { lmin = sv; lfirst = pos[c]; lastU = un ? pos[c] : -1; }
          else if (sv == lmin) { lfirst = imin(lfirst, pos[c]); if (un) lastU = imax(lastU, pos[c]); }
        }
      }
      const double gmin = wave_min_f64_dpp(lmin);
      if (!(gmin < INFINITY)) { infeasible = true; break; }
      const bool tied = lfirst != INT_MAX && lmin == gmin;
      const int gfirst = wave_min_i32_dpp(tied ? lfirst : INT_MAX);
      const int gU = wave_max_i32_dpp(tied ? lastU : -1);
      const int index = gU >= 0 ? gU : gfirst;
      minVal = gmin;
      // the column at list position `index` (exactly one lane / slot holds it)
      int mine = -1;
#pragma unroll
      for (int c = 0; c < NCL; ++c)
        if (pos[c] == index) mine = c;
      const unsigned long long own = __ballot(mine >= 0);
      const int src = __builtin_ctzll(own);
      const int j = __builtin_amdgcn_readlane(lane + 64 * (mine < 0 ? 0 : mine), src);
      const int r4 = __builtin_amdgcn_readlane(mine < 0 ? -1 : r4c[mine], src);
      // remaining[index] = remaining[--num]: the column at the last position moves to `index`, the chosen one leaves
      --num;
#pragma unroll
      for (int c = 0; c < NCL; ++c) {
        if (pos[c] == num && num != index) pos[c] = index;
        else if (pos[c] == index && lane + 64 * c == j) { pos[c] = -1; sc[c] = true; }
      }
      if (r4 < 0) sink = j; else i = r4;
    }
    if (infeasible) break;
    // duals: u over the visited rows needs the path costs of their columns (other lanes' registers -> LDS)
#pragma unroll
    for (int c = 0; c < NCL; ++c)
      if (lane + 64 * c < nc) spc_l[lane + 64 * c] = spc[c];
    __syncthreads();
    for (int e = lane; e < nr; e += 64) {
      if (e == cur) u[e] += minVal;
      else if (SR[e]) u[e] += minVal - spc_l[col4row[e]];
    }
#pragma unroll
    for (int c = 0; c < NCL; ++c)
      if (sc[c]) v[c] -= minVal - spc[c];
    __syncthreads();
    // augment along the path: j -> i = path[j]; row4col[j] = i; swap(col4row[i], j)
    int j = sink;
    for (;;) {
      const int ow = j & 63, oc = j >> 6;
      const int ii = __builtin_amdgcn_readlane(oc == 0 ? path[0] : path[1], ow);
      if (lane == ow) { if (oc == 0) r4c[0] = ii; else r4c[1] = ii; }
      const int t = col4row[ii];
      __syncthreads();
      if (lane == 0) col4row[ii] = j;
      __syncthreads();
      j = t;
      if (ii == cur) break;
    }
  }
  if (infeasible) {
    if (lane == 0) status[blockIdx.x] = 1;
    for (int e = lane; e < nr; e += 64) { rout[e] = e; cout_[e] = e; }
    return;
  }
  if (lane == 0) status[blockIdx.x] = 0;
  if (!tr) {
    for (int e = lane; e < nr; e += 64) { rout[e] = e; cout_[e] = col4row[e]; }
  } else {
    for (int e = lane; e < nr; e += 64) {
      const int q = col4row[e];
      int rank = 0;
      for (int t = 0; t < nr; ++t) rank += col4row[t] < q ? 1 : 0;
      rout[rank] = q;
      cout_[rank] = e;
    }
  }
}

// one wave = one problem.  nr0 x nc0 = the caller's matrix; internally rows = the smaller side.
__global__ __launch_bounds__(64) void lsap_kernel(const float* __restrict__ cost_all, int nr0, int nc0, int stage,
                                                  int64_t* __restrict__ row_ind, int64_t* __restrict__ col_ind,
                                                  int32_t* __restrict__ status) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int lane = threadIdx.x;
  const bool tr = nr0 > nc0;
  const int nr = tr ? nc0 : nr0, nc = tr ? nr0 : nc0;
  const float* __restrict__ cost = cost_all + (int64_t)blockIdx.x * nr0 * nc0;
  const LsapLayout L = lsap_layout(nr, nc, stage != 0);
  double* u = (double*)(lds + L.u);
  double* v = (double*)(lds + L.v);
  double* spc = (double*)(lds + L.spc);
  int* path = (int*)(lds + L.path);
  int* row4col = (int*)(lds + L.row4col);
  int* remaining = (int*)(lds + L.remaining);
  int* col4row = (int*)(lds + L.col4row);
  unsigned char* SR = lds + L.SR;
  unsigned char* SC = lds + L.SC;
  float* cm = (float*)(lds + L.cm);
  const int m = nr;                                        // assignments per problem
  int64_t* rout = row_ind + (int64_t)blockIdx.x * m;
  int64_t* cout_ = col_ind + (int64_t)blockIdx.x * m;

  for (int e = lane; e < nr; e += 64) { u[e] = 0.0; col4row[e] = -1; }
  for (int e = lane; e < nc; e += 64) { v[e] = 0.0; row4col[e] = -1; path[e] = -1; }
  bool invalid = false;                                    // NaN / -inf anywhere: scipy's "invalid numeric entries"
  for (int e = lane; e < nr0 * nc0; e += 64) { const float x = cost[e]; invalid |= (x != x) || x == -INFINITY; }
  if (stage) {
    for (int e = lane; e < nr * nc; e += 64) {
      const int i = e / nc, j = e - i * nc;
      cm[e] = tr ? cost[(int64_t)j * nc0 + i] : cost[(int64_t)i * nc0 + j];
    }
  }
  __syncthreads();
  bool infeasible = __ballot(invalid) != 0ull;
  for (int cur = 0; cur < nr && !infeasible; ++cur) {
    double minVal = 0.0;
    int num = nc;
    for (int e = lane; e < nc; e += 64) { remaining[e] = nc - e - 1; spc[e] = INFINITY; SC[e] = 0; }
    for (int e = lane; e < nr; e += 64) SR[e] = 0;
    __syncthreads();
    int i = cur, sink = -1;
    while (sink < 0) {
      if (lane == 0) SR[i] = 1;
      const double ui = u[i];
      double lmin = INFINITY;
      int lfirst = INT_MAX, lastU = -1;
      for (int it = lane; it < num; it += 64) {
        const int j = remaining[it];
        const float c = stage ? cm[i * nc + j] : (tr ? cost[(int64_t)j * nc0 + i] : cost[(int64_t)i * nc0 + j]);
        const double r = ((minVal + (double)c) - ui) - v[j];
        double s = spc[j];
        if (r < s) { path[j] = i; spc[j] = r; s = r; }
        const bool un = row4col[j] < 0;
        if (s < lmin) { lmin = s; lfirst = it; lastU = un ? it : -1; }
        else if (s == lmin && un) lastU = it;
      }
      const double gmin = wave_min_f64(lmin);
      if (!(gmin < INFINITY)) { infeasible = true; break; }     // all-infinite or NaN costs: scipy raises ValueError
      const bool tied = lfirst != INT_MAX && lmin == gmin;
      const int gfirst = wave_min_i32(tied ? lfirst : INT_MAX);
      const int gU = wave_max_i32(tied ? lastU : -1);
      const int index = gU >= 0 ? gU : gfirst;
      minVal = gmin;
      const int j = remaining[index];
      const int r4 = row4col[j];
      __syncthreads();                                  // every lane has read the list before it is edited
      if (lane == 0) { SC[j] = 1; remaining[index] = remaining[num - 1]; }
      --num;
      if (r4 < 0) sink = j; else i = r4;
      __syncthreads();
    }
    if (infeasible) break;
    for (int e = lane; e < nr; e += 64) {
      if (e == cur) u[e] += minVal;
      else if (SR[e]) u[e] += minVal - spc[col4row[e]];
    }
    for (int e = lane; e < nc; e += 64)
      if (SC[e]) v[e] -= minVal - spc[e];
    __syncthreads();
    if (lane == 0) {                                     // augment along the path (a handful of steps)
      int j = sink;
      for (;;) {
        const int ii = path[j];
        row4col[j] = ii;
        const int t = col4row[ii];
        col4row[ii] = j;
        j = t;
        if (ii == cur) break;
      }
    }
    __syncthreads();
  }
  if (infeasible) {                                      // flagged; the indices stay in range (identity)
    if (lane == 0) status[blockIdx.x] = 1;
    for (int e = lane; e < m; e += 64) { rout[e] = e; cout_[e] = e; }
    return;
  }
  if (lane == 0) status[blockIdx.x] = 0;
  if (!tr) {
    for (int e = lane; e < nr; e += 64) { rout[e] = e; cout_[e] = col4row[e]; }
  } else {                                               // scipy: rows ascending = argsort of col4row
    for (int e = lane; e < nr; e += 64) {
      const int q = col4row[e];
      int rank = 0;
      for (int t = 0; t < nr; ++t) rank += col4row[t] < q ? 1 : 0;
      rout[rank] = q;
      cout_[rank] = e;
    }
  }
}

}  // namespace
}  // namespace usc

using namespace usc;

extern "C" {

int usc_lsap_batch(const float* cost, int32_t n_prob, int32_t nr, int32_t nc, int64_t* row_ind, int64_t* col_ind,
                   int32_t* status, usc_stream_t s) {
  USC_REQUIRE(n_prob >= 0 && nr >= 0 && nc >= 0, "usc_lsap_batch: negative size");
  if (n_prob == 0 || nr == 0 || nc == 0) return USC_OK;
  USC_REQUIRE(cost && row_ind && col_ind && status, "usc_lsap_batch: null argument");
  const int r = nr < nc ? nr : nc, c = nr < nc ? nc : nr;
  bool stage = true;
  LsapLayout L = lsap_layout(r, c, true);
  if (L.total > kLsapLdsBudget) { stage = false; L = lsap_layout(r, c, false); }
  USC_REQUIRE(L.total <= kLsapLdsBudget, "usc_lsap_batch: %d x %d is too large for the one-wave solver (LDS)", nr, nc);
  static const bool lds_form = getenv("USC3D_LSAP_LDS") != nullptr;      // A/B switch: the LDS-resident form
  if (c <= 128 && stage && !lds_form)
    hipLaunchKernelGGL(lsap_reg_kernel, dim3(n_prob), dim3(64), (size_t)L.total, as_stream(s), cost, (int)nr, (int)nc,
                       row_ind, col_ind, status);
  else
    hipLaunchKernelGGL(lsap_kernel, dim3(n_prob), dim3(64), (size_t)L.total, as_stream(s), cost, (int)nr, (int)nc,
                       stage ? 1 : 0, row_ind, col_ind, status);
  USC_CHECK_LAUNCH("usc_lsap_batch");
  return USC_OK;
}

}  // extern "C"

// ===================================================================================================
// The set criterion itself (reference models/matcher.py:98-168 cost matrices; models/criterion.py:22-73 dice / BCE,
// :138-216 label + mask losses) for all prediction levels of one scene in a handful of launches:
//   crit_partial   per (level, 32-row chunk): sum_s softplus(x), sum_s sigmoid(x), and per target t
//                  sum_s x*tm[t,s], sum_s sigmoid(x)*tm[t,s]           (x = mask logits [S, Q], one query per lane)
//   crit_cost      chunks summed in order -> cost_mask = (sum softplus - sum x*tm)/S  (== mean BCE: softplus(-x) =
//                  softplus(x) - x), cost_dice = 1 - (2 N + 1)/(sum sigmoid + |tm| + 1), cost_class = -softmax[label]
//                  -> C = w_mask*cost_mask + w_class*cost_class + w_dice*cost_dice  [L, Q, T]
//   usc_lsap_batch the assignment (above)
//   crit_loss      the matcher's cost_mask / cost_dice ARE the mask losses of a (query, target) pair: the loss of a
//                  level is their sum over the matched pairs / T; weighted cross entropy over the queries
//   crit_bwd_*     d loss / d mask logits (matched columns only, everything else exactly 0) and d loss / d class logits
// Every sum has a fixed order; the target masks travel as one bit per (row, target) (T <= 32).
namespace usc {
namespace {

constexpr int kCritMaxLevels = 16;
constexpr int kCritCols = 128;          // queries per level (one per lane of the 128-thread workgroups)
struct CritLevels { const float* x[kCritMaxLevels]; float* dx[kCritMaxLevels]; };

__device__ inline float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ inline float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// tm u8[T, S] -> bits u32[S] (bit t = tm[t, s] != 0), cnt i32[T] += popcount (integer atomics: exact)
__global__ __launch_bounds__(256) void crit_target_bits_kernel(const uint8_t* __restrict__ tm, int T, int S,
                                                               uint32_t* __restrict__ bits, int32_t* __restrict__ cnt) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  uint32_t b = 0;
  if (s < S)
    for (int t = 0; t < T; ++t) b |= (tm[(int64_t)t * S + s] != 0 ? 1u : 0u) << t;
  if (s < S) bits[s] = b;
  for (int t = 0; t < T; ++t) {
    const unsigned long long m = __ballot((b >> t) & 1u);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&cnt[t], (int)__popcll(m));
  }
}

template <int TMAX>
__global__ __launch_bounds__(128) void crit_partial_kernel(CritLevels lv, int ld, int S, int Q,
                                                           const uint32_t* __restrict__ bits, int nchunk,
                                                           float* __restrict__ partial) {
  const int q = threadIdx.x, chunk = blockIdx.x, l = blockIdx.y;
  const float* __restrict__ X = lv.x[l];
  float nsum = 0.f, ssum = 0.f, xs[TMAX], gs[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) { xs[t] = 0.f; gs[t] = 0.f; }
  const int s0 = chunk * 32, s1 = s0 + 32 < S ? s0 + 32 : S;
  if (q < Q) {
    for (int s = s0; s < s1; ++s) {
      const float x = X[(int64_t)s * ld + q];
      const uint32_t b = bits[s];
      const float sp = softplus_f(x), sg = sigmoid_f(x);
      nsum += sp;
      ssum += sg;
#pragma unroll
      for (int t = 0; t < TMAX; ++t) {
        const bool m = (b >> t) & 1u;
        xs[t] += m ? x : 0.f;
        gs[t] += m ? sg : 0.f;
      }
    }
  }
  float* dst = partial + ((int64_t)(l * nchunk + chunk) * (2 * TMAX + 2)) * kCritCols + q;
  dst[0] = nsum;
  dst[kCritCols] = ssum;
#pragma unroll
  for (int t = 0; t < TMAX; ++t) { dst[(2 + t) * kCritCols] = xs[t]; dst[(2 + TMAX + t) * kCritCols] = gs[t]; }
}

struct CritCostArgs {
  const float* partial; int nchunk;
  const float* logits; int64_t ls_level, ls_q; int C;          // logits[l*ls_level + q*ls_q + c]
  const int64_t* labels; const int32_t* cnt;
  int S, Q, T; float w_mask, w_class, w_dice;
  float* cost;      // [L, Q, T]  (the LSAP input)
  float* cmask;     // [L, Q, T]
  float* cdice;     // [L, Q, T]
  float* nmat;      // [L, Q, T]  sum_s sigmoid(x) tm
  float* ssum;      // [L, Q]     sum_s sigmoid(x)
  float* logp;      // [L, Q, C]  log softmax of the class logits
};

template <int TMAX>
__global__ __launch_bounds__(128) void crit_cost_kernel(CritCostArgs a) {
  const int q = threadIdx.x, l = blockIdx.x;
  if (q >= a.Q) return;
  float nsum = 0.f, ssum = 0.f, xs[TMAX], gs[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) { xs[t] = 0.f; gs[t] = 0.f; }
  for (int c = 0; c < a.nchunk; ++c) {
    const float* src = a.partial + ((int64_t)(l * a.nchunk + c) * (2 * TMAX + 2)) * kCritCols + q;
    nsum += src[0];
    ssum += src[kCritCols];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) { xs[t] += src[(2 + t) * kCritCols]; gs[t] += src[(2 + TMAX + t) * kCritCols]; }
  }
  // class part: log softmax over the C logits of this query
  const float* lg = a.logits + (int64_t)l * a.ls_level + (int64_t)q * a.ls_q;
  float mx = -INFINITY;
  for (int c = 0; c < a.C; ++c) mx = fmaxf(mx, lg[c]);
  float se = 0.f;
  for (int c = 0; c < a.C; ++c) se += expf(lg[c] - mx);
  const float lse = logf(se);
  float* lp = a.logp + ((int64_t)l * a.Q + q) * a.C;
  for (int c = 0; c < a.C; ++c) lp[c] = (lg[c] - mx) - lse;
  a.ssum[(int64_t)l * a.Q + q] = ssum;
  const int64_t o = ((int64_t)l * a.Q + q) * a.T;
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    if (t < a.T) {
      const int64_t lab = a.labels[t];
      // a label outside [0, C) that is not the ignore id: no read past the logits; the NaN cost makes the assignment
      // report status 1, which the host side raises on (scipy raises ValueError on such a matrix, matcher.py:163)
      const bool bad = lab != 253 && (lab < 0 || lab >= a.C);
      const float cclass = lab == 253 ? -1.f : bad ? NAN : -(expf(lg[lab] - mx) / se);
      const float cm = (nsum - xs[t]) / (float)a.S;
      const float cd = 1.f - (2.f * gs[t] + 1.f) / (ssum + (float)a.cnt[t] + 1.f);
      a.cmask[o + t] = cm;
      a.cdice[o + t] = cd;
      a.nmat[o + t] = gs[t];
      a.cost[o + t] = (a.w_mask * cm + a.w_class * cclass) + a.w_dice * cd;
    }
  }
}

// one workgroup per level: losses of the matched pairs + weighted cross entropy
__global__ __launch_bounds__(128) void crit_loss_kernel(const float* __restrict__ cmask, const float* __restrict__ cdice,
                                                        const float* __restrict__ logp, const int64_t* __restrict__ src,
                                                        const int64_t* __restrict__ tid, const int64_t* __restrict__ labels,
                                                        const float* __restrict__ class_w, int Q, int T, int C, int noobj,
                                                        int32_t* __restrict__ tcls, float* __restrict__ part) {
  __shared__ int tc[kCritCols];
  __shared__ float rnum[kCritCols], rden[kCritCols];
  const int q = threadIdx.x, l = blockIdx.x;
  tc[q] = noobj;
  __syncthreads();
  if (q < T) tc[(int)src[(int64_t)l * T + q]] = (int)labels[tid[(int64_t)l * T + q]];
  __syncthreads();
  float num = 0.f, den = 0.f;
  if (q < Q) {
    int c = tc[q];
    if (c != 253 && (c < 0 || c >= C)) { c = 253; num = NAN; }     // invalid label: NaN loss, no out-of-range read
    tcls[(int64_t)l * Q + q] = c;
    if (c != 253) {
      const float w = class_w[c];
      num = -logp[((int64_t)l * Q + q) * C + c] * w;
      den = w;
    }
  }
  rnum[q] = num;
  rden[q] = den;
  __syncthreads();
  for (int o = kCritCols / 2; o > 0; o >>= 1) {          // fixed pairwise tree
    if (q < o) { rnum[q] += rnum[q + o]; rden[q] += rden[q + o]; }
    __syncthreads();
  }
  if (q == 0) {
    float lm = 0.f, ldice = 0.f;
    for (int t = 0; t < T; ++t) {
      const int64_t o = ((int64_t)l * Q + src[(int64_t)l * T + t]) * T + tid[(int64_t)l * T + t];
      lm += cmask[o];
      ldice += cdice[o];
    }
    part[l * 4 + 0] = rnum[0];
    part[l * 4 + 1] = rden[0];
    part[l * 4 + 2] = lm / (float)T;
    part[l * 4 + 3] = ldice / (float)T;
  }
}

// parts [B, L, 4] -> table [L, 4] = (sum_b num / sum_b den, sum_b mask, sum_b dice, 0), den_tot [L]
__global__ __launch_bounds__(64) void crit_table_kernel(const float* __restrict__ parts, int B, int L,
                                                        float* __restrict__ table, float* __restrict__ den_tot) {
  const int l = blockIdx.x * 64 + threadIdx.x;
  if (l >= L) return;
  float num = 0.f, den = 0.f, lm = 0.f, ld = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* p = parts + ((int64_t)b * L + l) * 4;
    num += p[0]; den += p[1]; lm += p[2]; ld += p[3];
  }
  table[l * 4 + 0] = num / den;
  table[l * 4 + 1] = lm;
  table[l * 4 + 2] = ld;
  table[l * 4 + 3] = 0.f;
  den_tot[l] = den;
}

struct CritBwdArgs {
  CritLevels lv; int ld, S, Q, T, nchunk;
  const uint32_t* bits; const int32_t* cnt;
  const int64_t* src; const int64_t* tid;
  const float* nmat; const float* ssum;
  const float* gtable;      // d total / d table [L, 4]
};

// d loss / d mask logits of every level: grid (chunk, level), one column per thread, full padded width written
__global__ __launch_bounds__(128) void crit_bwd_masks_kernel(CritBwdArgs a) {
  __shared__ int qmap[kCritCols];
  const int col = threadIdx.x, chunk = blockIdx.x, l = blockIdx.y;
  qmap[col] = -1;
  __syncthreads();
  if (col < a.T) qmap[(int)a.src[(int64_t)l * a.T + col]] = (int)a.tid[(int64_t)l * a.T + col];
  __syncthreads();
  if (col >= a.ld) return;
  const int t = col < a.Q ? qmap[col] : -1;
  const float* __restrict__ X = a.lv.x[l];
  float* __restrict__ dX = a.lv.dx[l];
  const int s0 = chunk * 32, s1 = s0 + 32 < a.S ? s0 + 32 : a.S;
  if (t < 0) {
    for (int s = s0; s < s1; ++s) dX[(int64_t)s * a.ld + col] = 0.f;
    return;
  }
  const float N = a.nmat[((int64_t)l * a.Q + col) * a.T + t];
  const float D1 = a.ssum[(int64_t)l * a.Q + col] + (float)a.cnt[t] + 1.f;
  const float ga = a.gtable[l * 4 + 1] / ((float)a.S * (float)a.T);
  const float gb = a.gtable[l * 4 + 2] / (float)a.T / (D1 * D1);
  const float c1 = 2.f * N + 1.f, c2 = 2.f * D1;
  for (int s = s0; s < s1; ++s) {
    const float x = X[(int64_t)s * a.ld + col];
    const float y = (a.bits[s] >> t) & 1u ? 1.f : 0.f;
    const float sg = sigmoid_f(x);
    dX[(int64_t)s * a.ld + col] = ga * (sg - y) + gb * (sg * (1.f - sg)) * (c1 - y * c2);
  }
}

// d loss / d class logits: dlogits[l*ls_level + q*ls_q + c] for one scene
__global__ __launch_bounds__(128) void crit_bwd_logits_kernel(const float* __restrict__ logp, const int32_t* __restrict__ tcls,
                                                              const float* __restrict__ class_w,
                                                              const float* __restrict__ gtable,
                                                              const float* __restrict__ den_tot, int Q, int C,
                                                              int64_t ls_level, int64_t ls_q, float* __restrict__ dlogits) {
  const int q = threadIdx.x, l = blockIdx.x;
  if (q >= Q) return;
  const int tc = tcls[(int64_t)l * Q + q];
  const float w = tc == 253 ? 0.f : class_w[tc];
  const float g = gtable[l * 4 + 0] * w / den_tot[l];
  const float* lp = logp + ((int64_t)l * Q + q) * C;
  float* d = dlogits + (int64_t)l * ls_level + (int64_t)q * ls_q;
  for (int c = 0; c < C; ++c) d[c] = g * (expf(lp[c]) - (c == tc ? 1.f : 0.f));
}

template <int TMAX>
static void launch_cost(const CritLevels& lv, int L, int ld, int S, int Q, const uint32_t* bits, int nchunk,
                        float* partial, const CritCostArgs& ca, hipStream_t st) {
  hipLaunchKernelGGL(crit_partial_kernel<TMAX>, dim3(nchunk, L), dim3(128), 0, st, lv, ld, S, Q, bits, nchunk, partial);
  hipLaunchKernelGGL(crit_cost_kernel<TMAX>, dim3(L), dim3(128), 0, st, ca);
}

static int tmax_of(int T) { return T <= 8 ? 8 : (T <= 16 ? 16 : 32); }

}  // namespace
}  // namespace usc

extern "C" {

int64_t usc_criterion_ws_bytes(int32_t L, int32_t S, int32_t T) {
  const int64_t nchunk = usc::ceil_div(S, 32);
  return usc::align_up((int64_t)L * nchunk * (2 * usc::tmax_of(T) + 2) * usc::kCritCols * 4, 256);
}

int usc_criterion_target_bits(const uint8_t* tm, int32_t T, int32_t S, uint32_t* bits, int32_t* cnt, usc_stream_t s) {
  USC_REQUIRE(T >= 1 && T <= 32 && S >= 1, "usc_criterion_target_bits: needs 1..32 targets");
  USC_REQUIRE(tm && bits && cnt, "usc_criterion_target_bits: null argument");
  hipStream_t st = usc::as_stream(s);
  (void)hipMemsetAsync(cnt, 0, (size_t)T * 4, st);
  hipLaunchKernelGGL(usc::crit_target_bits_kernel, dim3((unsigned)usc::ceil_div(S, 256)), dim3(256), 0, st, tm, (int)T,
                     (int)S, bits, cnt);
  USC_CHECK_LAUNCH("usc_criterion_target_bits");
  return USC_OK;
}

int usc_criterion_costs(const float* const* masks, int32_t L, int32_t ld, int32_t S, int32_t Q, int32_t T,
                        const uint32_t* bits, const int32_t* cnt, const float* logits, int64_t ls_level, int64_t ls_q,
                        int32_t C, const int64_t* labels, float w_mask, float w_class, float w_dice, float* cost,
                        float* cmask, float* cdice, float* nmat, float* ssum, float* logp, void* ws, int64_t ws_bytes,
                        usc_stream_t s) {
  using namespace usc;
  USC_REQUIRE(L >= 1 && L <= kCritMaxLevels && Q >= 1 && Q <= kCritCols && ld >= Q && T >= 1 && T <= 32 && S >= 1 && C >= 1,
              "usc_criterion_costs: needs <= 16 levels, <= 128 queries, 1..32 targets");
  USC_REQUIRE(masks && bits && cnt && logits && labels && cost && cmask && cdice && nmat && ssum && logp && ws &&
                  ws_bytes >= usc_criterion_ws_bytes(L, S, T), "usc_criterion_costs: bad argument");
  CritLevels lv{};
  for (int l = 0; l < L; ++l) { USC_REQUIRE(masks[l], "usc_criterion_costs: null level"); lv.x[l] = masks[l]; }
  const int nchunk = (int)ceil_div(S, 32);
  CritCostArgs ca{(const float*)ws, nchunk, logits, ls_level, ls_q, C, labels, cnt, S, Q, T, w_mask, w_class, w_dice,
                  cost, cmask, cdice, nmat, ssum, logp};
  hipStream_t st = as_stream(s);
  switch (tmax_of(T)) {
    case 8: launch_cost<8>(lv, L, ld, S, Q, bits, nchunk, (float*)ws, ca, st); break;
    case 16: launch_cost<16>(lv, L, ld, S, Q, bits, nchunk, (float*)ws, ca, st); break;
    default: launch_cost<32>(lv, L, ld, S, Q, bits, nchunk, (float*)ws, ca, st); break;
  }
  USC_CHECK_LAUNCH("usc_criterion_costs");
  return USC_OK;
}

int usc_criterion_losses(const float* cmask, const float* cdice, const float* logp, const int64_t* src, const int64_t* tid,
                         const int64_t* labels, const float* class_w, int32_t L, int32_t Q, int32_t T, int32_t C,
                         int32_t noobj, int32_t* tcls, float* part, usc_stream_t s) {
  using namespace usc;
  USC_REQUIRE(L >= 1 && Q >= 1 && Q <= kCritCols && T >= 1 && T <= 32 && T <= Q && C >= 1 && noobj >= 0 && noobj < C,
              "usc_criterion_losses: bad sizes");
  USC_REQUIRE(cmask && cdice && logp && src && tid && labels && class_w && tcls && part, "usc_criterion_losses: null argument");
  hipLaunchKernelGGL(crit_loss_kernel, dim3(L), dim3(128), 0, as_stream(s), cmask, cdice, logp, src, tid, labels, class_w,
                     (int)Q, (int)T, (int)C, (int)noobj, tcls, part);
  USC_CHECK_LAUNCH("usc_criterion_losses");
  return USC_OK;
}

int usc_criterion_table(const float* parts, int32_t B, int32_t L, float* table, float* den_tot, usc_stream_t s) {
  using namespace usc;
  USC_REQUIRE(B >= 1 && L >= 1 && parts && table && den_tot, "usc_criterion_table: bad argument");
  hipLaunchKernelGGL(crit_table_kernel, dim3((unsigned)ceil_div(L, 64)), dim3(64), 0, as_stream(s), parts, (int)B, (int)L,
                     table, den_tot);
  USC_CHECK_LAUNCH("usc_criterion_table");
  return USC_OK;
}

int usc_criterion_backward(const float* const* masks, float* const* dmasks, int32_t L, int32_t ld, int32_t S, int32_t Q,
                           int32_t T, const uint32_t* bits, const int32_t* cnt, const int64_t* src, const int64_t* tid,
                           const float* nmat, const float* ssum, const float* logp, const int32_t* tcls,
                           const float* class_w, const float* gtable, const float* den_tot, int32_t C, int64_t ls_level,
                           int64_t ls_q, float* dlogits, usc_stream_t s) {
  using namespace usc;
  USC_REQUIRE(L >= 1 && L <= kCritMaxLevels && Q >= 1 && Q <= kCritCols && ld >= Q && ld <= kCritCols && T >= 1 &&
                  T <= 32 && S >= 1 && C >= 1, "usc_criterion_backward: bad sizes");
  USC_REQUIRE(masks && dmasks && bits && cnt && src && tid && nmat && ssum && logp && tcls && class_w && gtable &&
                  den_tot && dlogits, "usc_criterion_backward: null argument");
  CritBwdArgs a{};
  for (int l = 0; l < L; ++l) {
    USC_REQUIRE(masks[l] && dmasks[l], "usc_criterion_backward: null level");
    a.lv.x[l] = masks[l];
    a.lv.dx[l] = dmasks[l];
  }
  a.ld = ld; a.S = S; a.Q = Q; a.T = T; a.nchunk = (int)ceil_div(S, 32);
  a.bits = bits; a.cnt = cnt; a.src = src; a.tid = tid; a.nmat = nmat; a.ssum = ssum; a.gtable = gtable;
  hipStream_t st = as_stream(s);
  hipLaunchKernelGGL(crit_bwd_masks_kernel, dim3(a.nchunk, L), dim3(128), 0, st, a);
  hipLaunchKernelGGL(crit_bwd_logits_kernel, dim3(L), dim3(128), 0, st, logp, tcls, class_w, gtable, den_tot, (int)Q, (int)C,
                     ls_level, ls_q, dlogits);
  USC_CHECK_LAUNCH("usc_criterion_backward");
  return USC_OK;
}

}  // extern "C"
