// ncut.hip — masked Normalized-Cut pseudo masks on the device (SURVEY.md §8a rows N2-N4;
// reference pseudo_masks/unscene3d_pseudo_main.py:82-153, utils/freemask_utils.py:8-18).
//
// The reference builds a binary segment affinity matrix on the host (numpy) and calls
// scipy.linalg.eigh(D-A, D, subset_by_index=[1,2]) -> LAPACK dsygvx: Cholesky of the diagonal D,
// reduction to C = D^-1/2 (D-A) D^-1/2 from the LOWER triangle, Householder tridiagonalisation
// (dsytrd, uplo='L'), bisection for eigenvalue #2 (dstebz), inverse iteration (dstein: eigenvector
// of T scaled so that its largest-magnitude component is positive), back-transformation with the
// reflectors (dormtr) and x = D^-1/2 y.  The SIGN of that vector decides which side of the cut is
// "foreground" downstream (get_salient_areas, argmax seed), so the same sequence of orthogonal
// transformations is restated here in fp64 — the result matches LAPACK's vector including its sign
// (verified against scipy on the golden fixtures).  Everything is S x S (S = 600..3000 segments):
// latency-bound; the trailing-matrix symv / rank-2 update of every Householder step run chip-wide.
#include <stdlib.h>

#include "common.h"

namespace usc {

// ---------------------------------------------------------------------------
// similarity
__global__ __launch_bounds__(256) void ncut_rownorm_kernel(const float* __restrict__ F, const uint8_t* __restrict__ zero_rows,
                                                          int64_t S, int d, int cosine_again, float* __restrict__ out) {
  // F.normalize(p=2, eps=1e-12) [+ cosine_sim's own x / (||x|| + 1e-9)]; one wave per row.
  // zero_rows: the rows get_masked_affinity_matrix has multiplied by (1 - painting) = 0 (reference :122-135) — the
  // product is formed here (0 * f, sign and NaN behaviour included) instead of by a pass over the features per iteration
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= S) return;
  const float keep = (zero_rows && zero_rows[r]) ? 0.f : 1.f;
  float ss = 0.f;
  for (int c = lane; c < d; c += 64) { const float v = keep * F[r * d + c]; ss += v * v; }
  ss = wave_reduce_addf(ss);
  const float n1 = fmaxf(sqrtf(ss), 1e-12f);
  float ss2 = 0.f;
  for (int c = lane; c < d; c += 64) { const float v = keep * F[r * d + c] / n1; ss2 += v * v; }
  ss2 = wave_reduce_addf(ss2);
  const float n2 = sqrtf(ss2) + 1e-9f;
  for (int c = lane; c < d; c += 64) {
    float v = keep * F[r * d + c] / n1;
    if (cosine_again) v = v / n2;
    out[r * d + c] = v;
  }
}

// sim[i][j] = sum_c X[i][c] * X[j][c]   (fp32, 16x16 register-free tile per block through LDS)
__global__ __launch_bounds__(256) void ncut_gram_kernel(const float* __restrict__ X, int64_t S, int d,
                                                       float* __restrict__ sim) {
  __shared__ float a[16][33], b[16][33];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.y * 16 + ty, j = (int64_t)blockIdx.x * 16 + tx;
  float acc = 0.f;
  for (int c0 = 0; c0 < d; c0 += 32) {
    for (int q = threadIdx.x; q < 16 * 32; q += 256) {
      const int r = q >> 5, c = q & 31;
      const int64_t ri = (int64_t)blockIdx.y * 16 + r, rj = (int64_t)blockIdx.x * 16 + r;
      a[r][c] = (ri < S && c0 + c < d) ? X[ri * d + c0 + c] : 0.f;
      b[r][c] = (rj < S && c0 + c < d) ? X[rj * d + c0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 32; ++c) acc = fmaf(a[ty][c], b[tx][c], acc);
    __syncthreads();
  }
  if (i < S && j < S) sim[i * S + j] = acc;
}

// cosine_sim's per-row min-max: attn -= rowmin; attn /= rowmax(after) + 1e-9
__global__ __launch_bounds__(256) void ncut_row_minmax_kernel(float* __restrict__ sim, int64_t S) {
  __shared__ float smin[4], smax[4];
  const int64_t r = blockIdx.x;
  float mn = 3.4e38f, mx = -3.4e38f;
  for (int64_t c = threadIdx.x; c < S; c += 256) { const float v = sim[r * S + c]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
  __syncthreads();
  mn = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
  mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  const float den = (mx - mn) + 1e-9f;
  for (int64_t c = threadIdx.x; c < S; c += 256) sim[r * S + c] = (sim[r * S + c] - mn) / den;
}

// normalize_mat: A -= min(A[A != 0]) if any(A > 0); A[A<0] = 0; A /= A.max() + 1e-5
// pass 1: block partials of (min over nonzero, any positive, max)
__global__ __launch_bounds__(256) void ncut_normmat_reduce_kernel(const float* __restrict__ A, int64_t numel,
                                                                 float* __restrict__ part /*[blocks][3]*/) {
  __shared__ float s0[4], s1[4], s2[4];
  float mn = 3.4e38f, mx = -3.4e38f, pos = 0.f;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < numel; q += (int64_t)gridDim.x * 256) {
    const float v = A[q];
    if (v != 0.f) mn = fminf(mn, v);
    if (v > 0.f) pos = 1.f;
    mx = fmaxf(mx, v);
  }
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); pos = fmaxf(pos, __shfl_xor(pos, o, 64));
  }
  if ((threadIdx.x & 63) == 0) { s0[threadIdx.x >> 6] = mn; s1[threadIdx.x >> 6] = mx; s2[threadIdx.x >> 6] = pos; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 3 + 0] = fminf(fminf(s0[0], s0[1]), fminf(s0[2], s0[3]));
    part[blockIdx.x * 3 + 1] = fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3]));
    part[blockIdx.x * 3 + 2] = fmaxf(fmaxf(s2[0], s2[1]), fmaxf(s2[2], s2[3]));
  }
}
__global__ __launch_bounds__(256) void ncut_normmat_apply_kernel(float* __restrict__ A, int64_t numel,
                                                                const float* __restrict__ part, int nparts) {
  float mn = 3.4e38f, mx = -3.4e38f, pos = 0.f;
  for (int b = 0; b < nparts; ++b) { mn = fminf(mn, part[b * 3]); mx = fmaxf(mx, part[b * 3 + 1]); pos = fmaxf(pos, part[b * 3 + 2]); }
  const float sub = pos > 0.f ? mn : 0.f;
  // max after the subtraction and the clip: max(mx - sub, 0)
  const float den = fmaxf(mx - sub, 0.f) + 1e-5f;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < numel; q += (int64_t)gridDim.x * 256) {
    float v = A[q] - sub;
    if (v < 0.f) v = 0.f;
    A[q] = v / den;
  }
}

// A = ((simA [+ simB]) / (1 or 2)) > tau, painted rows/cols forced off; deg[j] = sum_i (A_ij ? 1 : eps)
__global__ __launch_bounds__(256) void ncut_binarize_kernel(const float* __restrict__ simA, const float* __restrict__ simB,
                                                           int64_t S, float tau, const uint8_t* __restrict__ painted,
                                                           uint8_t* __restrict__ Abin) {
  const int64_t numel = S * S;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < numel; q += (int64_t)gridDim.x * 256) {
    float v = simA[q];
    if (simB) v = (v + simB[q]) / 2.f;
    uint8_t on = v > tau;
    if (painted) {
      const int64_t i = q / S, j = q - i * S;
      if (painted[i] || painted[j]) on = 0;
    }
    Abin[q] = on;
  }
}
__global__ __launch_bounds__(256) void ncut_degree_kernel(const float* __restrict__ simA, const float* __restrict__ simB,
                                                         int64_t S, float tau, double eps, double* __restrict__ deg) {
  // D is computed BEFORE the painted rows/cols are overwritten (reference :111-118 vs :426-427)
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= S) return;
  // rows first to last, one f64 add per row (numpy's order for A.sum(0)); the loads of 16 rows are issued together —
  // one row per iteration made every add wait for an L2 round trip (198 us for 625 rows)
  double s = 0.0;
  constexpr int U = 16;
  int64_t i = 0;
  for (; i + U <= S; i += U) {
    float va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      va[u] = simA[(i + u) * S + j];
      vb[u] = simB ? simB[(i + u) * S + j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float v = simB ? (va[u] + vb[u]) / 2.f : va[u];
      s += (v > tau) ? 1.0 : eps;
    }
  }
  for (; i < S; ++i) {
    float v = simA[i * S + j];
    if (simB) v = (v + simB[i * S + j]) / 2.f;
    s += (v > tau) ? 1.0 : eps;
  }
  deg[j] = s;
}

// C = D^-1/2 (D - A) D^-1/2 from the LOWER triangle of A (mirrored), full symmetric storage
__global__ __launch_bounds__(256) void ncut_laplacian_kernel(const uint8_t* __restrict__ Abin, const double* __restrict__ deg,
                                                            int64_t S, double eps, double* __restrict__ C) {
  const int64_t numel = S * S;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < numel; q += (int64_t)gridDim.x * 256) {
    const int64_t i = q / S, j = q - i * S;
    const int64_t lo = i >= j ? i * S + j : j * S + i;
    const double a = Abin[lo] ? 1.0 : eps;
    const double v = ((i == j) ? deg[i] : 0.0) - a;
    C[q] = v / (sqrt(deg[i]) * sqrt(deg[j]));
  }
}

// ---------------------------------------------------------------------------
// Householder tridiagonalisation, LAPACK dsytd2 uplo='L' conventions.
// step i:  (1) reflector from column i  (2) p = C22 v  (3) C22 -= v w^T + w v^T, w = tau p - (tau^2/2)(p.v) v
// The two tridiagonalisation paths must round alike, and C must stay exactly symmetric (the stepwise path reads
// column i where the one-launch path reads row i): the rank-2 update is written once, both products rounded before
// they are added — v_r w_c + w_r v_c then commutes bit for bit, which a fused multiply-add would break.
__device__ __forceinline__ double tri_w(double tau, double p, double a2, double v) {
  const double t = a2 * v;
  return __builtin_fma(tau, p, t);
}
__device__ __forceinline__ double tri_update(double x, double vr, double wc, double wr, double vc) {
  const double t1 = vr * wc;
  const double t2 = wr * vc;
  const double u = t1 + t2;
  return x - u;
}

struct TriState {
  double* C;      // [n][n] symmetric working matrix
  double* Vt;     // [n][n] row i = reflector i (v[i+1] = 1, zeros above)
  double* d;      // [n]
  double* e;      // [n-1]
  double* tau;    // [n-1]
  double* p;      // [n] scratch
  int64_t n;
};

// The sixteen wave sums of a reflector's squared norm, added as a balanced tree (four dependent adds instead of
// fifteen), and the Householder scalars from them.  ||(alpha, x)|| is one square root of alpha^2 + |x|^2 — the
// entries of a normalised Laplacian are <= 2 in magnitude, so the squares cannot overflow; sqrt(tot) followed by hypot
// cost ~60 dependent f64 instructions on the critical path of every step.  Shared by the stepwise and the one-launch
// kernels: both give the same T and reflectors bit for bit.
__device__ inline double sum16_tree(const double* red) {
  const double a0 = red[0] + red[1], a1 = red[2] + red[3], a2 = red[4] + red[5], a3 = red[6] + red[7];
  const double a4 = red[8] + red[9], a5 = red[10] + red[11], a6 = red[12] + red[13], a7 = red[14] + red[15];
  return ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}
__device__ inline void householder_scalars(double alpha, double tot, double& beta, double& tau, double& scale) {
  if (tot == 0.0) {
    beta = alpha; tau = 0.0; scale = 0.0;
  } else {
    const double nrm = sqrt(alpha * alpha + tot);
    beta = alpha >= 0.0 ? -nrm : nrm;     // -sign(alpha) * ||(alpha, x)||
    tau = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
  }
}

// Steps (1) + (2) in one launch: every workgroup recomputes the reflector of column i — bit for bit the reduction of
// tri_reflector_kernel's single 1024-thread workgroup (each of the 256 threads plays four of its threads, the sixteen
// wave sums are added in the same order) — keeps v in LDS and then forms its rows of p = C22 v, one wave per row.
// Workgroup 0 also stores d, e, tau and v (for the update kernel and the back-transform).  Two launches per
// Householder step instead of three: the solver is bound by the ~5 us kernel-to-kernel latency of its dependent
// chain (1 875 launches for 625 segments), not by work.
__global__ __launch_bounds__(256) void tri_reflect_symv_kernel(TriState t, int64_t i) {
  extern __shared__ double vsh[];   // [n]
  __shared__ double red[16];
  __shared__ double s_scale;
  const int64_t n = t.n;
  const int tid = threadIdx.x, lane = tid & 63;
  const double* col = t.C;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    double ss = 0.0;
    for (int64_t r = i + 2 + tid + 256 * v; r < n; r += 1024) { const double x = col[r * n + i]; ss += x * x; }
    ss = wave_reduce_addd(ss);
    if (lane == 0) red[(tid >> 6) + 4 * v] = ss;
  }
  __syncthreads();
  if (tid == 0) {
    const double tot = sum16_tree(red);
    const double alpha = col[(i + 1) * n + i];
    double beta, tau, scale;
    householder_scalars(alpha, tot, beta, tau, scale);
    s_scale = scale;
    if (blockIdx.x == 0) {
      t.d[i] = col[i * n + i];
      t.e[i] = beta;
      t.tau[i] = tau;
      if (i == n - 2) t.d[n - 1] = 0.0;  // set by the final step
    }
  }
  __syncthreads();
  const double scale = s_scale;
  for (int64_t r = tid; r < n; r += 256) {
    double val = 0.0;
    if (r == i + 1) val = 1.0;
    else if (r > i + 1) val = col[r * n + i] * scale;
    vsh[r] = val;
    if (blockIdx.x == 0) t.Vt[i * n + r] = val;
  }
  __syncthreads();
  const int64_t r = i + 1 + (int64_t)blockIdx.x * 4 + (tid >> 6);
  if (r >= n) return;
  const double* row = t.C + r * n;
  double s = 0.0;
  for (int64_t c = i + 1 + lane; c < n; c += 64) s += row[c] * vsh[c];
  s = wave_reduce_addd(s);
  if (lane == 0) t.p[r] = s;
}

// C22 -= v w^T + w v^T with w = tau p + a2 v, a2 = -tau^2/2 (p.v); each block recomputes the scalar
__global__ __launch_bounds__(256) void tri_update_kernel(TriState t, int64_t i) {
  __shared__ double red[4];
  const int64_t n = t.n;
  const double tau = t.tau[i];
  if (tau == 0.0) return;
  const double* v = t.Vt + i * n;
  double dot = 0.0;
  for (int64_t c = i + 1 + threadIdx.x; c < n; c += 256) dot += t.p[c] * v[c];
  dot = wave_reduce_addd(dot);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
  __syncthreads();
  dot = red[0] + red[1] + red[2] + red[3];
  const double a2 = -0.5 * tau * (tau * dot);
  const int64_t m = n - i - 1;
  const int64_t total = m * m;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
    const int64_t r = i + 1 + q / m, c = i + 1 + q % m;
    const double wr = tri_w(tau, t.p[r], a2, v[r]), wc = tri_w(tau, t.p[c], a2, v[c]);
    t.C[r * n + c] = tri_update(t.C[r * n + c], v[r], wc, wr, v[c]);
  }
}
// ---------------------------------------------------------------------------
// The whole tridiagonalisation in ONE launch.  The two-launch-per-step path above spends its time in the ~5 us
// kernel-to-kernel latency of 2(n-1) dependent launches; here G co-resident workgroups keep the working matrix
// distributed by rows (row r lives with workgroup r mod G, in LDS when it fits) and meet at one grid barrier per
// Householder step (the exchange of p below):
//   before the exchange of step i every workgroup builds reflector i from the pivot column it already holds
//                                 (redundantly — same reduction order as tri_reflect_symv_kernel), forms its rows of
//                                 p = C22 v, and the owner of row i+1 publishes that row as it stands;
//   after the exchange            every workgroup has p, forms w = tau p - (tau^2/2)(p.v) v, applies the rank-2
//                                 update to its own rows and — to the published row i+1 — redundantly, which is the
//                                 pivot column of step i+1 (C stays exactly symmetric: v_r w_c + w_r v_c commutes).
// p and the published row travel as 16-byte {value, step tag} slots, stored write-through and polled by the threads
// that need them: the data is its own arrival flag, so a step costs one store -> load trip through memory and no
// counter, drain or fence.  Slots ping-pong between two buffers (nobody can run more than one step ahead).
// The arithmetic per element is that of the stepwise kernels, so both paths give the same T and reflectors.
// Waits are bounded: a thread that does not see its slots arrive raises the error word and everybody leaves
// (d[0] becomes NaN, which the caller sees as a NaN eigenpair) instead of hanging the device.
struct Slot {          // 16 bytes, written and read as ONE dwordx4 access: a value never shows without its tag
  double value;
  uint64_t tag;
};
struct TriPersist {
  TriState t;
  Slot* pbuf;        // [2][n]  p of step i lives in pbuf[i & 1], tagged i + 1
  Slot* rowbuf;      // [2][n]  row i+1 as it stood before the update of step i
  unsigned int* sync;  // [1] error flag   (zeroed before launch, with the slots)
  int G;
  int R;             // rows per workgroup
  int ld;            // LDS row pitch (doubles)
};
constexpr unsigned int kSpinLimit = 1u << 20;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// write-through / cache-bypassing 16-byte buffer accesses (aux = sc1): visible across the eight XCD L2s without
// fences; as compiler intrinsics (not asm) several loads can be in flight before one wait
constexpr int kSc1 = 16;
__device__ inline u32x4 slot_pack(double value, uint64_t tag) {
  u32x4 q;
  q.x = (unsigned int)__double2loint(value); q.y = (unsigned int)__double2hiint(value);
  q.z = (unsigned int)tag; q.w = (unsigned int)(tag >> 32);
  return q;
}
__device__ inline double slot_value(const u32x4& q) { return __hiloint2double((int)q.y, (int)q.x); }
__device__ inline uint64_t slot_tag(const u32x4& q) { return (uint64_t)q.z | ((uint64_t)q.w << 32); }

// Reductions of doubles on the DPP path (row shifts and row broadcasts: VALU latency) instead of ds_bpermute
// (__shfl_xor: two LDS-crossbar trips per level) — a Householder step is a chain of dependent reductions.
// The DPP sequence adds neighbours first (lane bit 0, then 1, ... 5); the xor butterfly of wave_reduce_addd adds
// distance 32 first (bit 5, then 4, ... 0).  Both are balanced trees, so with the summands placed in bit-reversed lane
// order the DPP tree is the butterfly's tree and the sums are bit-identical to the stepwise kernels'.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_f64(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
// sum over each row of 16 lanes; valid in the row's last lane (15, 31, 47, 63)
__device__ inline double row16_sum(double x) {
  x += dpp_f64<0x111, 0xf>(x);   // row_shr:1
  x += dpp_f64<0x112, 0xf>(x);   // row_shr:2
  x += dpp_f64<0x114, 0xf>(x);   // row_shr:4
  x += dpp_f64<0x118, 0xf>(x);   // row_shr:8
  return x;
}
__device__ inline int bitrev6(int l) { return (int)(__brev((unsigned int)l) >> 26); }
__device__ inline int bitrev4(int l) { return (int)(__brev((unsigned int)l) >> 28); }
// sum over the wave, broadcast to every lane
__device__ inline double wave_sum_dpp(double x) {
  x = row16_sum(x);
  x += dpp_f64<0x142, 0xa>(x);   // row_bcast:15 into rows 1 and 3
  x += dpp_f64<0x143, 0xc>(x);   // row_bcast:31 into rows 2 and 3
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
  return __hiloint2double(hi, lo);
}

// ROWS: where a workgroup keeps its rows of the working matrix.
//   kRowsReg    registers: wave w holds the workgroup's rows q = w, w+4, ... (RW of them), lane e the columns e + 64 j
//               (J of them) — symv and the rank-2 update then run on registers.  The stepwise kernel's lane e' sums the
//               columns i+1+e' (mod 64), so lane e's partial sum belongs at tree position e - (i+1) mod 64: one
//               ds_bpermute rotates it there (bit-reversed, see wave_sum_dpp) and the sums stay bit-identical.
//   kRowsLds    LDS, 16 lanes per row (n too large for registers);  kRowsGlobal  global memory (too large for LDS).
enum { kRowsReg = 0, kRowsLds = 1, kRowsGlobal = 2 };
template <int ROWS, int J, int RW>
__global__ __launch_bounds__(256) void tri_persistent_kernel(TriPersist a) {
  constexpr bool LDS_ROWS = ROWS == kRowsLds;
  extern __shared__ double lds[];
  __shared__ double red[20];
  __shared__ int s_abort;
  const TriState& t = a.t;
  const int64_t n = t.n;
  const int G = a.G, R = a.R, g = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int erev = bitrev6(lane), frev = bitrev4(tid & 15);
  const int64_t ld = a.ld;   // row pitch in LDS: = 16 mod 32 doubles, so two 16-lane row groups hit disjoint banks
  const __amdgpu_buffer_rsrc_t slots = __builtin_amdgcn_make_buffer_rsrc((void*)a.pbuf, 0, (int)(4 * n * 16), 0x00020000);
  double* v = lds;            // [n] reflector
  double* w = v + n;          // [n]
  double* p = w + n;          // [n]
  double* piv = p + n;        // [n] pivot column of the current step (entries >= i valid)
  double* rows_l = piv + n;   // [R][n] when LDS_ROWS
  auto row_ptr = [&](int q) -> double* {
    return LDS_ROWS ? rows_l + (int64_t)q * ld : t.C + ((int64_t)g + (int64_t)q * G) * n;
  };
  if (LDS_ROWS) {
    for (int q = 0; q < R; ++q) {
      const int64_t r = g + (int64_t)q * G;
      if (r < n) for (int64_t c = tid; c < n; c += 256) rows_l[(int64_t)q * ld + c] = t.C[r * n + c];
    }
  }
  double rowreg[RW][J];
  if (ROWS == kRowsReg) {
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      const int64_t r = g + (int64_t)(wave + 4 * k) * G;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int64_t c = lane + 64 * j;
        rowreg[k][j] = (wave + 4 * k < R && r < n && c < n) ? t.C[r * n + c] : 0.0;
      }
    }
  }
  for (int64_t c = tid; c < n; c += 256) piv[c] = t.C[c];   // row 0 == column 0
  if (tid == 0) s_abort = 0;
  __syncthreads();

#ifdef USC_TRI_TIMING   // developer build (-DUSC_TRI_TIMING): cycles per phase of workgroups 0 and G-1, printed at the end
  long long tA = 0, tB = 0, tC = 0, tD = 0, t0 = clock64(), t1;
#define TRI_MARK(acc) do { t1 = clock64(); acc += t1 - t0; t0 = t1; } while (0)
#else
#define TRI_MARK(acc) do {} while (0)
#endif
  for (int64_t i = 0; i + 1 < n; ++i) {
    // ---- reflector i from piv (x = piv[i+2..], alpha = piv[i+1]); every thread ends up with tau and scale ----
    {
      // the stepwise kernel's 1024 virtual threads: group wave + 4u, position bitrev6(lane) (see above)
      double ss[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ss[u] = 0.0;
        for (int64_t r = i + 2 + 64 * (wave + 4 * u) + erev; r < n; r += 1024) { const double x = piv[r]; ss[u] += x * x; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) ss[u] = wave_sum_dpp(ss[u]);
      if (lane == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) red[wave + 4 * u] = ss[u];
      }
    }
    __syncthreads();
    double tau, scale;
    {
      const double tot = sum16_tree(red);
      const double alpha = piv[i + 1];
      double beta;
      householder_scalars(alpha, tot, beta, tau, scale);
      if (g == 0 && tid == 0) {
        t.d[i] = piv[i];
        t.e[i] = beta;
        t.tau[i] = tau;
      }
    }
    for (int64_t r = tid; r < n; r += 256) {
      double val = 0.0;
      if (r == i + 1) val = 1.0;
      else if (r > i + 1) val = piv[r] * scale;
      v[r] = val;
      if (g == 0) t.Vt[i * n + r] = val;
    }
    __syncthreads();
    TRI_MARK(tA);
    if (i + 2 >= n) break;   // last step: tau = 0 (empty x), nothing to update
    // ---- own rows of p = C22 v; the owner of row i+1 publishes it ----
    // slots of step i: p in [par*n, par*n + n), the published row in [(2 + par)*n, ...), par = i & 1
    const int pb_off = (int)((i & 1) * n) * 16, rb_off = (int)((2 + (i & 1)) * n) * 16;
    const uint64_t tag = (uint64_t)i + 1;
    if (ROWS == kRowsReg) {
      double vreg[J], part[RW];
#pragma unroll
      for (int j = 0; j < J; ++j) { const int64_t c = lane + 64 * j; vreg[j] = (c > i && c < n) ? v[c] : 0.0; }
#pragma unroll
      for (int k = 0; k < RW; ++k) {
        part[k] = 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j)
          if (lane + 64 * j > i && lane + 64 * j < n) part[k] += rowreg[k][j] * vreg[j];
      }
      const int src = ((erev + (int)((i + 1) & 63)) & 63) * 4;   // lane that holds tree position bitrev6(lane)
#pragma unroll
      for (int k = 0; k < RW; ++k) {
        const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(part[k]));
        const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(part[k]));
        part[k] = wave_sum_dpp(__hiloint2double(hi, lo));
      }
#pragma unroll
      for (int k = 0; k < RW; ++k) {
        const int64_t r = g + (int64_t)(wave + 4 * k) * G;
        if (lane == 0 && wave + 4 * k < R && r > i && r < n)
          __builtin_amdgcn_raw_buffer_store_b128(slot_pack(part[k], tag), slots, pb_off + (int)r * 16, 0, kSc1);
      }
      if ((i + 1) % G == g) {
        const int qo = (int)((i + 1) / G);
        if ((qo & 3) == wave) {
#pragma unroll
          for (int k = 0; k < RW; ++k)
            if (k == (qo >> 2)) {
#pragma unroll
              for (int j = 0; j < J; ++j) {
                const int64_t c = lane + 64 * j;
                if (c > i && c < n)
                  __builtin_amdgcn_raw_buffer_store_b128(slot_pack(rowreg[k][j], tag), slots, rb_off + (int)c * 16, 0, kSc1);
              }
            }
        }
      }
    } else {
      // 16 lanes per row, 16 rows per pass; a lane carries four of the stepwise kernel's 64 per-lane partial sums
      // (positions frev + 16 m) and folds them in the butterfly's order before the 16-lane DPP tree
      for (int q = tid >> 4; q < R; q += 16) {
        const int64_t r = g + (int64_t)q * G;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        if (r > i && r < n) {
          const double* row = row_ptr(q);
          // the four chains advance together (each keeps its own ascending-c order)
          for (int64_t c0 = i + 1 + frev; c0 < n; c0 += 64) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const int64_t c = c0 + 16 * m;
              if (c < n) acc[m] += row[c] * v[c];
            }
          }
        }
        double sum = (acc[0] + acc[2]) + (acc[1] + acc[3]);
        sum = row16_sum(sum);
        if ((tid & 15) == 15 && r > i && r < n)
          __builtin_amdgcn_raw_buffer_store_b128(slot_pack(sum, tag), slots, pb_off + (int)r * 16, 0, kSc1);
      }
      if ((i + 1) % G == g) {
        const double* row = row_ptr((int)((i + 1) / G));
        for (int64_t c = i + 1 + tid; c < n; c += 256)
          __builtin_amdgcn_raw_buffer_store_b128(slot_pack(row[c], tag), slots, rb_off + (int)c * 16, 0, kSc1);
      }
    }
    // ---- exchange: every thread waits for the tagged slots it needs (this is the grid barrier: nobody gets past
    // step i without everybody's p of step i).  No counter, no drain, no atomic; a slot is rewritten at step i+2, by
    // which time every workgroup has produced step i+1 and therefore finished reading step i ----
    TRI_MARK(tB);
    constexpr int K = 4;   // slots per thread and buffer in flight together
    for (int64_t c0 = i + 1 + tid; c0 < n; c0 += 256 * K) {
      u32x4 qp[K], qr[K];
      unsigned int spins = 0;
      for (;;) {
        asm volatile("" ::: "memory");   // the slots change under us: reload every round
        bool ok = true;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int64_t c = c0 + 256 * k;
          if (c < n) {
            qp[k] = __builtin_amdgcn_raw_buffer_load_b128(slots, pb_off + (int)c * 16, 0, kSc1);
            qr[k] = __builtin_amdgcn_raw_buffer_load_b128(slots, rb_off + (int)c * 16, 0, kSc1);
          }
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
          if (c0 + 256 * k < n) ok = ok && slot_tag(qp[k]) == tag && slot_tag(qr[k]) == tag;
        if (ok) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinLimit || __hip_atomic_load(&a.sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(&a.sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_abort = 1;
          break;
        }
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int64_t c = c0 + 256 * k;
        if (c < n) { p[c] = slot_value(qp[k]); piv[c] = slot_value(qr[k]); }
      }
    }
    __syncthreads();
    TRI_MARK(tC);
    if (s_abort) {
      if (g == 0 && tid == 0) t.d[0] = __builtin_nan("");
      return;
    }
    double dot = 0.0;
    for (int64_t c = i + 1 + 64 * wave + erev; c < n; c += 256) dot += p[c] * v[c];
    dot = wave_sum_dpp(dot);
    if (lane == 0) red[16 + wave] = dot;
    __syncthreads();
    dot = red[16] + red[17] + red[18] + red[19];
    const double a2 = -0.5 * tau * (tau * dot);
    for (int64_t c = i + 1 + tid; c < n; c += 256) w[c] = tri_w(tau, p[c], a2, v[c]);
    __syncthreads();
    if (tau != 0.0 && ROWS == kRowsReg) {
      double vreg[J], wreg[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int64_t c = lane + 64 * j;
        vreg[j] = (c > i && c < n) ? v[c] : 0.0;
        wreg[j] = (c > i && c < n) ? w[c] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < RW; ++k) {
        const int64_t r = g + (int64_t)(wave + 4 * k) * G;
        if (wave + 4 * k < R && r > i && r < n) {
          const double vr = v[r], wr = w[r];
#pragma unroll
          for (int j = 0; j < J; ++j)
            if (lane + 64 * j > i && lane + 64 * j < n) rowreg[k][j] = tri_update(rowreg[k][j], vr, wreg[j], wr, vreg[j]);
        }
      }
    } else if (tau != 0.0) {
      for (int q = tid >> 4; q < R; q += 16) {
        const int64_t r = g + (int64_t)q * G;
        if (r <= i || r >= n) continue;
        double* row = row_ptr(q);
        const double vr = v[r], wr = w[r];
        // four columns per trip, loads first: a load behind a store to `row` would wait for it (same address space)
        for (int64_t c0 = i + 1 + (tid & 15); c0 < n; c0 += 64) {
          double x[4], wc[4], vc[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int64_t c = c0 + 16 * m;
            if (c < n) { x[m] = row[c]; wc[m] = w[c]; vc[m] = v[c]; }
          }
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int64_t c = c0 + 16 * m;
            if (c < n) row[c] = tri_update(x[m], vr, wc[m], wr, vc[m]);
          }
        }
      }
    }
    {
      const double vr = v[i + 1], wr = w[i + 1];
      for (int64_t c = i + 1 + tid; c < n; c += 256) {
        double x = piv[c];
        if (tau != 0.0) x = tri_update(x, vr, w[c], wr, v[c]);
        piv[c] = x;
      }
    }
    __syncthreads();
    TRI_MARK(tD);
  }
#ifdef USC_TRI_TIMING
  if ((g == 0 || g == G - 1) && tid == 0)
    printf("tri timing wg %d: reflector %lld  symv+publish %lld  exchange %lld  update %lld cycles (n=%d)\n", g, tA, tB, tC, tD, (int)n);
#endif
#undef TRI_MARK
  // d[n-1]: bottom-right entry after the last update; its owner holds it
  if ((n - 1) % G == g) {
    const int qo = (int)((n - 1) / G);
    if (ROWS == kRowsReg) {
      if ((qo & 3) == wave && lane == (int)((n - 1) & 63)) {
#pragma unroll
        for (int k = 0; k < RW; ++k)
#pragma unroll
          for (int j = 0; j < J; ++j)
            if (k == (qo >> 2) && j == (int)((n - 1) >> 6)) t.d[n - 1] = rowreg[k][j];
      }
    } else if (tid == 0) {
      t.d[n - 1] = row_ptr(qo)[n - 1];
    }
  }
}

__global__ void tri_last_diag_kernel(TriState t) {
  if (threadIdx.x == 0 && blockIdx.x == 0) t.d[t.n - 1] = t.C[(t.n - 1) * t.n + (t.n - 1)];
}

// ---------------------------------------------------------------------------
// eigenvalue #k of the tridiagonal (Sturm counts, 64 shifts per round) + inverse iteration with partial
// pivoting; the vector is scaled to unit norm with its largest-magnitude component positive (dstein).
__device__ inline int sturm_count(const double* d, const double* e, int64_t n, double x) {
  int cnt = 0;
  double q = d[0] - x;
  if (q < 0.0) ++cnt;
  for (int64_t j = 1; j < n; ++j) {
    const double den = (q != 0.0) ? q : 1e-300;
    q = d[j] - x - e[j - 1] * e[j - 1] / den;
    if (q < 0.0) ++cnt;
  }
  return cnt;
}

// The same count from the determinant recurrence  p_j = (d_j - x) p_{j-1} - e_{j-1}^2 p_{j-2}  (q_j = p_j / p_{j-1}): the
// dependent chain of a step is ONE fma instead of an f64 division (~10 dependent instructions) — a bisection round over
// n = 625 drops from ~55 us to ~10 us.  p is rescaled by a power of two (exact; growth per step is
// bounded by |d - x| + e^2, shrinkage by cancellation); p rescaled every eight steps; a collapsed or non-finite chain
// sends the lane to the ratio form (neither happens on real Laplacians).
__device__ inline int sturm_count_fast(const double* d, const double* e, const double* e2, int64_t n, double x) {
  double p0 = 1.0, p1 = d[0] - x;
  unsigned sg = (unsigned)__double2hiint(p1) >> 31;   // sign bits of p, newest in bit 0
  int cnt = (int)sg;
  int64_t j = 1;
  double dj[8], ej[8];
  if (j + 8 <= n) {
#pragma unroll
    for (int u = 0; u < 8; ++u) { dj[u] = d[j + u]; ej[u] = e2[j + u - 1]; }
  }
  for (; j + 8 <= n; j += 8) {
    double dc[8], ec[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { dc[u] = dj[u]; ec[u] = ej[u]; }
    if (j + 16 <= n) {      // operands of the next block: in flight during this block's chain
#pragma unroll
      for (int u = 0; u < 8; ++u) { dj[u] = d[j + 8 + u]; ej[u] = e2[j + 8 + u - 1]; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double p2 = (dc[u] - x) * p1 - ec[u] * p0;
      sg = __builtin_amdgcn_alignbit(sg, (unsigned)__double2hiint(p2), 31);   // (sg << 1) | sign(p2)
      p0 = p1; p1 = p2;
    }
    cnt += __popc((sg ^ (sg >> 1)) & 0xffu);      // sign changes among the last nine values
    int ex0, ex1;
    (void)frexp(p0, &ex0);
    (void)frexp(p1, &ex1);
    const int ex = ex0 > ex1 ? ex0 : ex1;
    p0 = ldexp(p0, -ex);
    p1 = ldexp(p1, -ex);
  }
  for (; j < n; ++j) {
    const double p2 = (d[j] - x) * p1 - e2[j - 1] * p0;
    cnt += (int)(((unsigned)__double2hiint(p2) ^ (unsigned)__double2hiint(p1)) >> 31);
    p0 = p1; p1 = p2;
  }
  // An exact zero in the middle of the chain gives the ratio form's count by itself (p_{j+1} = -e^2 p_{j-1}: one sign
  // change over the two steps either way); what the product form cannot survive is p_j = p_{j-1} = 0 (a split
  // matrix with x on an eigenvalue of the leading block) or an overflow — both stay visible at the end of the chain.
  if ((p1 == 0.0 && p0 == 0.0) || !(fabs(p1) < 1.7e308)) cnt = sturm_count(d, e, n, x);
  return cnt;
}

// LDS = 1: the tridiagonal (d, e), the LU work arrays and the iterate live in LDS (n <= kEigLdsMax) — the bisection and
// above all the single-lane recurrences of the inverse iteration are chains of dependent loads from global memory
// otherwise: 4.3 ms of a 12 ms solve.
// Round 4: every recurrence carries its state in registers and reads its operands eight steps at a time (a step used to
// be an LDS store -> load round trip, ~130 cycles; the chains are now an fma or two), the back substitution multiplies by
// reciprocals computed by all lanes, norm / arg-max / start vector are spread over the wave: 1.41 -> see DESIGN §5.
constexpr int64_t kEigLdsMax = 800;     // 10 n doubles of LDS (64 KB without an attribute)
template <int LDS>
__global__ __launch_bounds__(128) void tri_eig_kernel(const double* __restrict__ d_g, const double* __restrict__ e_g, int64_t n,
                                                    int k, double* __restrict__ eval_out /*[2]*/,
                                                    double* __restrict__ z_g /*[n]*/, double* __restrict__ work_g /*[7n]*/) {
  extern __shared__ double eig_sh[];   // LDS: d[n] e[n] z[n] work[5n] zprev[n] e2[n]
  // two waves: wave 0 bisects eigenvalue #k and goes straight on to its eigenvector, wave 1 bisects #k+1 (only
  // reported) at the same time
  const int lane = threadIdx.x & 63, which = threadIdx.x >> 6;
  const double* d = d_g;
  const double* e = e_g;
  double* z = z_g;
  double* work = work_g;
  double* zprev = work_g + 5 * n;      // (global form: the tridiagonalisation's exchange slots behind the work arrays, free by now)
  double* e2 = work_g + 6 * n;
  if (LDS) {
    double* dl_ = eig_sh;
    double* el_ = eig_sh + n;
    e2 = eig_sh + 9 * n;
    for (int64_t j = threadIdx.x; j < n; j += 128) {
      const double ej = j < n - 1 ? e_g[j] : 0.0;
      dl_[j] = d_g[j]; el_[j] = ej; e2[j] = ej * ej;
    }
    __syncthreads();
    d = dl_; e = el_; z = eig_sh + 2 * n; work = eig_sh + 3 * n; zprev = eig_sh + 8 * n;
  } else {
    for (int64_t j = threadIdx.x; j < n; j += 128) { const double ej = j < n - 1 ? e_g[j] : 0.0; e2[j] = ej * ej; }
    __syncthreads();
  }
#ifdef USC_EIG_TIMING   // developer build: cycles per phase of wave 0, printed at the end
  long long tph[6] = {0, 0, 0, 0, 0, 0}, tq0 = clock64(), tq1;
#define EIG_MARK(i) do { tq1 = clock64(); tph[i] += tq1 - tq0; tq0 = tq1; } while (0)
#else
#define EIG_MARK(i) do {} while (0)
#endif
  // Gershgorin interval
  double lo = 1e300, hi = -1e300;
  for (int64_t j = lane; j < n; j += 64) {
    const double r = (j > 0 ? fabs(e[j - 1]) : 0.0) + (j < n - 1 ? fabs(e[j]) : 0.0);
    lo = fmin(lo, d[j] - r);
    hi = fmax(hi, d[j] + r);
  }
  for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o, 64)); hi = fmax(hi, __shfl_xor(hi, o, 64)); }
  double lam;
  {
    double a = lo - 1e-12 - 1e-12 * fabs(lo), b = hi + 1e-12 + 1e-12 * fabs(hi);
    const int target = k + which;   // eigenvalue index (0-based): smallest x with count(x) > target
    for (int round = 0; round < 14; ++round) {
      const double x = a + (b - a) * (double)(lane + 1) / 65.0;
      const int cnt = sturm_count_fast(d, e, e2, n, x);
      const unsigned long long m = __ballot(cnt > target);   // lanes whose shift is above the eigenvalue
      const int first = m ? (__ffsll((long long)m) - 1) : 64;
      const double na = first == 0 ? a : a + (b - a) * (double)first / 65.0;
      const double nb = first == 64 ? b : a + (b - a) * (double)(first + 1) / 65.0;
      a = na; b = nb;
      // 65^9 > 2^53: the bracket reaches neighbouring doubles after nine rounds; further rounds cannot move it
      if (b - a <= 4.5e-16 * fmax(fabs(a), fabs(b))) break;      // (a, b are wave-uniform)
    }
    lam = 0.5 * (a + b);
  }
  if (lane == 0) eval_out[which] = lam;
  if (which != 0) return;
  EIG_MARK(0);
  // ---- inverse iteration on wave 0.  The LU sweep and the two substitutions are recurrences (lane 0); everything
  // element-wise (set-up, reciprocals, norm, the 1/norm scaling, arg-max, sign, copy-out) is spread over the 64 lanes.
  // Lane 0's LDS/global writes are ordered against the other lanes' later reads by program order within the wave plus
  // the waits below.
#define USC_WAVE_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
  double tnorm = 0.0;
  for (int64_t j = lane; j < n; j += 64)
    tnorm = fmax(tnorm, fabs(d[j]) + (j > 0 ? fabs(e[j - 1]) : 0.0) + (j < n - 1 ? fabs(e[j]) : 0.0));
  for (int o = 32; o > 0; o >>= 1) tnorm = fmax(tnorm, __shfl_xor(tnorm, o, 64));
  const double shift = lam;
  double* dl = work;            // sub-diagonal multipliers
  double* dd = work + n;        // U diagonal, then its reciprocal
  double* du = work + 2 * n;    // U first super-diagonal
  double* du2 = work + 3 * n;   // U second super-diagonal (from pivoting)
  double* piv = work + 4 * n;   // 1.0 if rows were swapped
  // LU with partial pivoting of T - shift*I (dgttrf's elimination order)
  const double tiny = 2.3e-16 * fmax(tnorm, 1e-300);
  for (int64_t j = lane; j < n; j += 64) { du2[j] = 0.0; dd[j] = d[j] - shift; }   // (du2[n-1] is read by the back substitution)
  // start vector: fixed pseudo-random in (-1,1): z[j] from state j+1 of  st <- a st + c;  lane L jumps L+1 steps, then 64
  {
    const unsigned long long la = 6364136223846793005ull, lc = 1442695040888963407ull;
    unsigned long long A = 1ull, Cc = 0ull;
    for (int t = 0; t < 64; ++t)
      if (t <= lane) { A = A * la; Cc = Cc * la + lc; }
    const unsigned long long A64 = __shfl(A, 63, 64), C64 = __shfl(Cc, 63, 64);
    unsigned long long st = A * 0x9E3779B97F4A7C15ull + Cc;
    for (int64_t j = lane; j < n; j += 64) {
      z[j] = ((double)(st >> 11) / 9007199254740992.0) * 2.0 - 1.0;
      st = A64 * st + C64;
    }
  }
  USC_WAVE_SYNC();
  EIG_MARK(1);
  if (lane == 0) {
    // state carried down the rows: U diagonal and first super-diagonal of the row being eliminated
    double ddj = dd[0], duj = n > 1 ? e[0] : 0.0;
    auto E = [&](int64_t j) { return (LDS || j < n - 1) ? e[j] : 0.0; };   // (the LDS copy stores e[n-1] = 0)
    auto lu_step = [&](int64_t j, double sub, double d_next, double e_next) {
      // sub = e[j];  d_next = d[j+1] - shift;  e_next = e[j+1] (0 past the end)
      const bool swap = !(fabs(ddj) >= fabs(sub));
      const double ddc = (!swap && fabs(ddj) < tiny) ? tiny : ddj;
      const double num = swap ? ddc : sub, den = swap ? sub : ddc;
      const double mlt = num / den;
      const double A = swap ? duj : d_next, B = swap ? d_next : duj;
      const bool fill = swap && j < n - 2;
      dl[j] = mlt; piv[j] = swap ? 1.0 : 0.0; dd[j] = den; du[j] = B; du2[j] = fill ? e_next : 0.0;
      ddj = A - mlt * B;
      duj = fill ? -mlt * e_next : e_next;
    };
    int64_t j = 0;
    for (; j + 8 <= n - 1; j += 8) {
      double sb[9], dn[8];
#pragma unroll
      for (int u = 0; u < 9; ++u) sb[u] = E(j + u);
#pragma unroll
      for (int u = 0; u < 8; ++u) dn[u] = dd[j + u + 1];     // d - shift, written by all lanes above
#pragma unroll
      for (int u = 0; u < 8; ++u) lu_step(j + u, sb[u], dn[u], sb[u + 1]);
    }
    for (; j < n - 1; ++j) lu_step(j, E(j), dd[j + 1], E(j + 1));
    if (fabs(ddj) < tiny) ddj = tiny;
    dd[n - 1] = ddj; du[n - 1] = 0.0;
  }
  USC_WAVE_SYNC();
  for (int64_t j = lane; j < n; j += 64) dd[j] = 1.0 / dd[j];
  USC_WAVE_SYNC();
  EIG_MARK(2);
  double prev_change = 0.0;
  for (int it = 0; it < 6; ++it) {
    if (lane == 0) {
      // forward: apply L^-1 with the recorded row swaps; zc = the entry carried down
      double zc = z[0];
      int64_t j = 0;
      for (; j + 8 <= n - 1; j += 8) {
        double zn[8], dlj[8], pj[8], out[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { zn[u] = z[j + u + 1]; dlj[u] = dl[j + u]; pj[u] = piv[j + u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool sw = pj[u] != 0.0;
          out[u] = sw ? zn[u] : zc;
          const double other = sw ? zc : zn[u];
          zc = other - dlj[u] * out[u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) z[j + u] = out[u];
      }
      for (; j < n - 1; ++j) {
        const double zn = z[j + 1], dlj = dl[j];
        const bool sw = piv[j] != 0.0;
        const double out = sw ? zn : zc, other = sw ? zc : zn;
        z[j] = out;
        zc = other - dlj * out;
      }
      z[n - 1] = zc;
      // backward: U x = z (dd holds the reciprocals)
      double z1 = z[n - 1] * dd[n - 1], z2 = 0.0;     // x[j+1], x[j+2]
      z[n - 1] = z1;
      j = n - 2;
      for (; j - 7 >= 0; j -= 8) {
        double zj[8], duj[8], du2j[8], rj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { zj[u] = z[j - u]; duj[u] = du[j - u]; du2j[u] = du2[j - u]; rj[u] = dd[j - u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double xj = (zj[u] - duj[u] * z1 - du2j[u] * z2) * rj[u];
          z2 = z1; z1 = xj; zj[u] = xj;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) z[j - u] = zj[u];
      }
      for (; j >= 0; --j) {
        const double xj = (z[j] - du[j] * z1 - du2[j] * z2) * dd[j];
        z2 = z1; z1 = xj; z[j] = xj;
      }
    }
    USC_WAVE_SYNC();
    EIG_MARK(3);
    double nrm = 0.0;
    for (int64_t j = lane; j < n; j += 64) nrm += z[j] * z[j];
    for (int o = 32; o > 0; o >>= 1) nrm += __shfl_xor(nrm, o, 64);
    nrm = sqrt(nrm);
    // normalise; the iteration has converged when the normalised iterate repeats (up to sign) to 1e-14 of its largest
    // entry — with the shift at the eigenvalue that takes two or three sweeps
    double dmax = 0.0, smax = 0.0, zmax = 0.0;
    for (int64_t j = lane; j < n; j += 64) {
      const double zn = z[j] / nrm, zp = zprev[j];
      dmax = fmax(dmax, fabs(zn - zp));
      smax = fmax(smax, fabs(zn + zp));
      zmax = fmax(zmax, fabs(zn));
      z[j] = zn;
      zprev[j] = zn;
    }
    for (int o = 32; o > 0; o >>= 1) {
      dmax = fmax(dmax, __shfl_xor(dmax, o, 64));
      smax = fmax(smax, __shfl_xor(smax, o, 64));
      zmax = fmax(zmax, __shfl_xor(zmax, o, 64));
    }
    USC_WAVE_SYNC();
    EIG_MARK(4);
    // (wave-uniform)  The shift is an eigenvalue to an ulp, so one sweep suppresses every other component by
    // ~1e-16 / gap: a change <= 1e-11 means the iterate BEFORE this sweep was already that close, and this sweep took it
    // to the noise floor (measured: 2.5e-13 after sweep 1, 9.4e-14 from then on — the old 1e-14 bar was never met
    // and all six sweeps ran).  Second exit: the change has stopped shrinking.
    const double change = fmin(dmax, smax);
    if (it >= 1 && (change <= 1e-11 * zmax || (it >= 2 && change >= 0.25 * prev_change))) break;
    prev_change = change;
  }
  // sign: the largest-magnitude component (first one on ties) is positive
  double best = -1.0;
  long long bj = 0;
  for (int64_t j = lane; j < n; j += 64) {
    const double a = fabs(z[j]);
    if (a > best) { best = a; bj = j; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double ob = __shfl_xor(best, o, 64);
    const long long oj = __shfl_xor(bj, o, 64);
    if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
  }
  const int flip = z[bj] < 0.0;
  for (int64_t j = lane; j < n; j += 64) {
    const double zj = flip ? -z[j] : z[j];
    if (LDS) z_g[j] = zj; else z[j] = zj;
  }
  EIG_MARK(5);
#ifdef USC_EIG_TIMING
  if (lane == 0)
    printf("eig timing: bisection %lld  setup %lld  lu %lld  sweeps %lld  normalise %lld  sign+copy %lld cycles (n=%d)\n", tph[0],
           tph[1], tph[2], tph[3], tph[4], tph[5], (int)n);
#endif
#undef EIG_MARK
#undef USC_WAVE_SYNC
}

// y = H(0) H(1) ... H(n-2) z, then x = y / sqrt(deg)
__global__ __launch_bounds__(1024) void tri_backtransform_kernel(TriState t, const double* __restrict__ deg,
                                                                double* __restrict__ z, double* __restrict__ x) {
  __shared__ double red[16];
  __shared__ double s_dot;
  const int64_t n = t.n;
  const int tid = threadIdx.x;
  for (int64_t i = n - 2; i >= 0; --i) {
    const double tau = t.tau[i];
    if (tau == 0.0) continue;   // block-uniform
    const double* v = t.Vt + i * n;
    double dot = 0.0;
    for (int64_t r = i + 1 + tid; r < n; r += 1024) dot += v[r] * z[r];
    dot = wave_reduce_addd(dot);
    if ((tid & 63) == 0) red[tid >> 6] = dot;
    __syncthreads();
    if (tid == 0) { double s = 0.0; for (int k = 0; k < 16; ++k) s += red[k]; s_dot = s; }
    __syncthreads();
    const double f = tau * s_dot;
    for (int64_t r = i + 1 + tid; r < n; r += 1024) z[r] -= f * v[r];
    __syncthreads();
  }
  for (int64_t r = tid; r < n; r += 1024) x[r] = z[r] / sqrt(deg[r]);
}

// The same on ONE wave for n <= 64 * PER (round 4): z lives in registers (element lane + 64 j), the next reflector's
// vector and tau are in flight while the current one is applied, the dot product is one wave reduction — no LDS, no
// workgroup barrier (the 1 024-thread form above spends ~1 us per reflector in three barriers and a global round trip
// of z: 0.66 ms of a 5.6 ms solve at n = 625; this form ~0.1 ms).  Same reflectors in the same order; the dot
// products are summed in a different order (last-bit differences in the eigenvector).
template <int PER>
__global__ __launch_bounds__(64) void tri_backtransform_wave_kernel(TriState t, const double* __restrict__ deg,
                                                                   const double* __restrict__ z, double* __restrict__ x) {
  const int lane = threadIdx.x;
  const int64_t n = t.n;
  double zr[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int64_t r = lane + 64 * j;
    zr[j] = r < n ? z[r] : 0.0;
  }
  // kBtDepth reflectors in flight (a reflector's 8 * n bytes come from L2 / HBM at ~1-2 us; applying one takes ~0.1 us)
  constexpr int D = PER <= 8 ? 8 : 6;
  double v[D][PER], tau[D];
  auto load = [&](int64_t i, double (&vv)[PER], double& tt) __attribute__((always_inline)) {
    if (i < 0) { tt = 0.0; return; }             // wave-uniform
    const double* vp = t.Vt + i * n;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int64_t r = lane + 64 * j;
      vv[j] = (r > i && r < n) ? vp[r] : 0.0;
    }
    tt = t.tau[i];
  };
  auto apply = [&](const double (&vv)[PER], double tt) __attribute__((always_inline)) {
    if (tt == 0.0) return;                       // wave-uniform
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < PER; ++j) dot += vv[j] * zr[j];
    dot = wave_sum_dpp(dot);                     // (ds_bpermute butterflies: 12 LDS-crossbar trips per reflector)
    const double f = tt * dot;
#pragma unroll
    for (int j = 0; j < PER; ++j) zr[j] -= f * vv[j];
  };
  int64_t i = n - 2;
#pragma unroll
  for (int q = 0; q < D; ++q) load(i - q, v[q], tau[q]);
  for (; i >= 0; i -= D) {
#pragma unroll
    for (int q = 0; q < D; ++q) {
      apply(v[q], tau[q]);
      load(i - D - q, v[q], tau[q]);
    }
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int64_t r = lane + 64 * j;
    if (r < n) x[r] = zr[j] / sqrt(deg[r]);
  }
}

// Four reflectors per step on four waves (round 4, second form).  The one-wave kernel above is bound by what ONE wave
// can keep in flight (the vmcnt counter stops at 63 loads = 33 KB; 3.1 MB of reflectors at ~2 us per round trip is
// 0.2-0.3 ms however short the arithmetic is).  Here wave w owns the elements r = lane + 64 (w + 4 j): four waves have
// four times the loads in flight, and the reflectors a = i, b = i-1, c = i-2, d = i-3 of a step are applied together —
//   f_a = tau_a (v_a.z)                      f_b = tau_b (v_b.z - f_a v_a.v_b)
//   f_c = tau_c (v_c.z - f_a v_a.v_c - f_b v_b.v_c)      f_d = ...            z -= f_a v_a + f_b v_b + f_c v_c + f_d v_d
// (what H(d) H(c) H(b) H(a) z is, with the intermediate z's eliminated) — so the ten dot products of a step share ONE
// reduction and one workgroup barrier instead of four dependent wave reductions.  Every wave forms the four factors
// from the same partial sums in the same order.  Differences to the sequential form: rounding only.
template <int PER>
__global__ __launch_bounds__(256) void tri_backtransform_quad_kernel(TriState t, const double* __restrict__ deg,
                                                                    const double* __restrict__ z, double* __restrict__ x) {
  constexpr int W = 4, B = 4, ND = 10;     // waves, reflectors per step, dot products per step
  constexpr int P = PER <= 2 ? 6 : 4;      // steps whose reflectors are in flight (P * B * PER loads per lane <= 60)
  __shared__ double part[2][ND][4 * W];   // [slot][dot product][wave * 4 + row of 16 lanes]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = t.n;
  double zr[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int64_t r = lane + 64 * (wave + W * j);
    zr[j] = r < n ? z[r] : 0.0;
  }
  double v[P][B][PER], tau[P][B];
  // reflectors are fetched strictly in descending order, one row pointer walking down Vt (a pointer per ring slot cost
  // more scalar registers than there are: the first build spilled ~600 readlane / writelane pairs per step)
  int rr[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) { rr[j] = lane + 64 * (wave + W * j); if (rr[j] >= (int)n) rr[j] = -1; }
  int inext = (int)n - 2;
  const double* pnext = t.Vt + (int64_t)inext * n;
  const double* tnext = t.tau + inext;
  auto load = [&](double (&vv)[B][PER], double (&tt)[B]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < B; ++q) {
      if (inext < 0) {                            // workgroup-uniform: past the first reflector
        tt[q] = 0.0;
#pragma unroll
        for (int j = 0; j < PER; ++j) vv[q][j] = 0.0;
      } else {
#pragma unroll
        for (int j = 0; j < PER; ++j) vv[q][j] = rr[j] > inext ? pnext[rr[j]] : 0.0;
        tt[q] = *tnext;
      }
      --inext; pnext -= n; --tnext;
    }
  };
  int step = 0;
  auto apply = [&](const double (&vv)[B][PER], const double (&tt)[B]) __attribute__((always_inline)) {
    // dots 0-3: v_q.z;  4-9: v_a.v_b, v_a.v_c, v_a.v_d, v_b.v_c, v_b.v_d, v_c.v_d
    double dsum[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) dsum[k] = 0.0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
#pragma unroll
      for (int q = 0; q < B; ++q) dsum[q] += vv[q][j] * zr[j];
      dsum[4] += vv[0][j] * vv[1][j]; dsum[5] += vv[0][j] * vv[2][j]; dsum[6] += vv[0][j] * vv[3][j];
      dsum[7] += vv[1][j] * vv[2][j]; dsum[8] += vv[1][j] * vv[3][j]; dsum[9] += vv[2][j] * vv[3][j];
    }
    // sums over the rows of 16 lanes (four DPP levels, no scalar round trip); the 4 x 4 row sums of a dot product meet in LDS
#pragma unroll
    for (int k = 0; k < ND; ++k) dsum[k] = row16_sum(dsum[k]);
    const int slot = step & 1;
    if ((lane & 15) == 15) {
#pragma unroll
      for (int k = 0; k < ND; ++k) part[slot][k][wave * 4 + (lane >> 4)] = dsum[k];
    }
    __syncthreads();
    // lane k < 10 adds the sixteen row sums of dot product k (balanced tree); the totals go round by readlane
    double mine = 0.0;
    if (lane < ND) {
      const double* pk = part[slot][lane];
      mine = (((pk[0] + pk[1]) + (pk[2] + pk[3])) + ((pk[4] + pk[5]) + (pk[6] + pk[7]))) +
             (((pk[8] + pk[9]) + (pk[10] + pk[11])) + ((pk[12] + pk[13]) + (pk[14] + pk[15])));
    }
    double dt[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k)
      dt[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mine), k), __builtin_amdgcn_readlane(__double2loint(mine), k));
    const double fa = tt[0] * dt[0];
    const double fb = tt[1] * (dt[1] - fa * dt[4]);
    const double fc = tt[2] * (dt[2] - fa * dt[5] - fb * dt[7]);
    const double fd = tt[3] * (dt[3] - fa * dt[6] - fb * dt[8] - fc * dt[9]);
#pragma unroll
    for (int j = 0; j < PER; ++j) zr[j] -= fa * vv[0][j] + fb * vv[1][j] + fc * vv[2][j] + fd * vv[3][j];
    ++step;
  };
#pragma unroll
  for (int q = 0; q < P; ++q) load(v[q], tau[q]);
  for (int i = (int)n - 2; i >= 0; i -= B * P) {
#pragma unroll
    for (int q = 0; q < P; ++q) {
      if (i - B * q >= 0) apply(v[q], tau[q]);     // workgroup-uniform
      load(v[q], tau[q]);
    }
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int64_t r = lane + 64 * (wave + W * j);
    if (r < n) x[r] = zr[j] / sqrt(deg[r]);
  }
}

// One-launch tridiagonalisation when the co-resident grid and its LDS fit; false -> caller runs the stepwise path.
// USC3D_TRI_STEPWISE=1 forces the stepwise path (A/B measurements).
static bool launch_persistent(const TriState& t, double* tail, hipStream_t st) {
  static const int mode = [] { const char* e = getenv("USC3D_TRI_STEPWISE"); return (e && e[0] == '1') ? 1 : 0; }();
  if (mode == 1) return false;
  const int64_t n = t.n;
  if (n < 8 || n > 4000) return false;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
    return false;
  TriPersist a;
  a.t = t;
  // few workgroups: every arrival is an atomic on one word, and a step's work is tiny (n = 625: 2.4 kFLOP per row)
  static const int g_env = [] { const char* e = getenv("USC3D_TRI_G"); return e ? atoi(e) : 0; }();
  a.G = g_env > 0 ? g_env : 64;
  if (a.G > cus) a.G = cus;
  if ((int64_t)a.G > n / 2) a.G = (int)(n / 2);
  a.R = (int)ceil_div(n, a.G);
  a.pbuf = reinterpret_cast<Slot*>(tail);
  a.rowbuf = a.pbuf + 2 * n;
  a.sync = reinterpret_cast<unsigned int*>(a.rowbuf + 2 * n);
  a.ld = (int)(((n + 15) / 32) * 32 + 16);   // >= n, = 16 mod 32
  const size_t vec_bytes = (size_t)4 * n * sizeof(double);
  const size_t row_bytes = (size_t)a.R * a.ld * sizeof(double);
  void (*kern)(TriPersist) = nullptr;
  size_t lds = vec_bytes;
  if (n <= 640 && a.R <= 12) {
    kern = tri_persistent_kernel<kRowsReg, 10, 3>;
  } else if (n <= 1024 && a.R <= 16) {
    kern = tri_persistent_kernel<kRowsReg, 16, 4>;
  } else if (vec_bytes + row_bytes <= 150 * 1024) {
    kern = tri_persistent_kernel<kRowsLds, 1, 1>;
    lds = vec_bytes + row_bytes;
  } else {
    kern = tri_persistent_kernel<kRowsGlobal, 1, 1>;
  }
  if (lds > 150 * 1024) return false;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return false;
  (void)hipMemsetAsync(tail, 0, (size_t)(8 * n + 2) * sizeof(double), st);   // tags 0 = "nothing published yet"
  hipLaunchKernelGGL(kern, dim3((unsigned)a.G), dim3(256), lds, st, a);
  return true;
}

}  // namespace usc

using namespace usc;

extern "C" {

int usc_ncut_similarity(const float* F, int64_t S, int32_t d, int32_t cosine_mode, float* normed, float* sim,
                        usc_stream_t s) {
  return usc_ncut_similarity_masked(F, nullptr, S, d, cosine_mode, normed, sim, s);
}

int usc_ncut_similarity_masked(const float* F, const uint8_t* zero_rows, int64_t S, int32_t d, int32_t cosine_mode,
                               float* normed, float* sim, usc_stream_t s) {
  USC_REQUIRE(S >= 1 && d >= 1 && F && normed && sim, "usc_ncut_similarity: bad argument");
  hipStream_t st = as_stream(s);
  hipLaunchKernelGGL(ncut_rownorm_kernel, dim3((unsigned)ceil_div(S, 4)), dim3(256), 0, st, F, zero_rows, S, (int)d,
                     (int)cosine_mode, normed);
  hipLaunchKernelGGL(ncut_gram_kernel, dim3((unsigned)ceil_div(S, 16), (unsigned)ceil_div(S, 16)), dim3(256), 0, st,
                     (const float*)normed, S, (int)d, sim);
  if (cosine_mode) hipLaunchKernelGGL(ncut_row_minmax_kernel, dim3((unsigned)S), dim3(256), 0, st, sim, S);
  USC_CHECK_LAUNCH("usc_ncut_similarity");
  return USC_OK;
}

int usc_ncut_normalize_mat(float* A, int64_t S, void* ws, int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(S >= 1 && A && ws && ws_bytes >= 1024 * 3 * 4, "usc_ncut_normalize_mat: bad argument");
  hipStream_t st = as_stream(s);
  const int64_t numel = S * S;
  int nb = (int)ceil_div(numel, 256 * 8);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(ncut_normmat_reduce_kernel, dim3(nb), dim3(256), 0, st, (const float*)A, numel, (float*)ws);
  hipLaunchKernelGGL(ncut_normmat_apply_kernel, dim3(stream_grid(numel, 256)), dim3(256), 0, st, A, numel,
                     (const float*)ws, nb);
  USC_CHECK_LAUNCH("usc_ncut_normalize_mat");
  return USC_OK;
}

int usc_ncut_binarize(const float* simA, const float* simB, int64_t S, float tau, double eps, const uint8_t* painted,
                      uint8_t* Abin, double* deg, usc_stream_t s) {
  USC_REQUIRE(S >= 1 && simA && Abin && deg, "usc_ncut_binarize: bad argument");
  hipStream_t st = as_stream(s);
  hipLaunchKernelGGL(ncut_degree_kernel, dim3((unsigned)ceil_div(S, 256)), dim3(256), 0, st, simA, simB, S, tau, eps, deg);
  hipLaunchKernelGGL(ncut_binarize_kernel, dim3(stream_grid(S * S, 256)), dim3(256), 0, st, simA, simB, S, tau, painted,
                     Abin);
  USC_CHECK_LAUNCH("usc_ncut_binarize");
  return USC_OK;
}

int64_t usc_ncut_fiedler_ws_bytes(int64_t S) { return (2 * S * S + 20 * S + 32) * 8; }

int usc_ncut_fiedler(const uint8_t* Abin, const double* deg, int64_t S, double eps, double* evec, double* eval,
                     void* ws, int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(S >= 3 && S <= 8000 && Abin && deg && evec && eval && ws, "usc_ncut_fiedler: bad argument (3 <= S <= 8000)");
  USC_REQUIRE(ws_bytes >= usc_ncut_fiedler_ws_bytes(S), "usc_ncut_fiedler: workspace too small");
  hipStream_t st = as_stream(s);
  double* w = (double*)ws;
  TriState t{};
  t.n = S;
  t.C = w;
  t.Vt = w + S * S;
  t.d = w + 2 * S * S;
  t.e = t.d + S;
  t.tau = t.e + S;
  t.p = t.tau + S;
  double* z = t.p + S;
  double* work = z + S;   // 5 S
  double* work_tail = work + 5 * S;   // offset 2S^2 + 10S doubles: 16-byte aligned; 8S + 2 doubles: tagged p / row slots + error word
  hipLaunchKernelGGL(ncut_laplacian_kernel, dim3(stream_grid(S * S, 256)), dim3(256), 0, st, Abin, deg, S, eps, t.C);
  if (!launch_persistent(t, work_tail, st)) {
    for (int64_t i = 0; i + 1 < S; ++i) {
      const int64_t m = S - i - 1;
      hipLaunchKernelGGL(tri_reflect_symv_kernel, dim3((unsigned)ceil_div(m, 4)), dim3(256), (size_t)S * sizeof(double), st, t, i);
      hipLaunchKernelGGL(tri_update_kernel, dim3(stream_grid(m * m, 256)), dim3(256), 0, st, t, i);
    }
    hipLaunchKernelGGL(tri_last_diag_kernel, dim3(1), dim3(64), 0, st, t);
  }
  if (S <= kEigLdsMax)
    hipLaunchKernelGGL(tri_eig_kernel<1>, dim3(1), dim3(128), (size_t)S * 10 * sizeof(double), st, (const double*)t.d,
                       (const double*)t.e, S, 1, eval, z, work);
  else
    hipLaunchKernelGGL(tri_eig_kernel<0>, dim3(1), dim3(128), 0, st, (const double*)t.d, (const double*)t.e, S, 1, eval, z, work);
  static const bool bt_wave = !(getenv("USC3D_BACKTRANSFORM_WAVE") && getenv("USC3D_BACKTRANSFORM_WAVE")[0] == '0');
  static const bool bt_quad = !(getenv("USC3D_BACKTRANSFORM_QUAD") && getenv("USC3D_BACKTRANSFORM_QUAD")[0] == '0');
  if (bt_quad && S <= 512)
    hipLaunchKernelGGL(tri_backtransform_quad_kernel<2>, dim3(1), dim3(256), 0, st, t, deg, (const double*)z, evec);
  else if (bt_quad && S <= 768)
    hipLaunchKernelGGL(tri_backtransform_quad_kernel<3>, dim3(1), dim3(256), 0, st, t, deg, (const double*)z, evec);
  else if (bt_quad && S <= 1024)
    hipLaunchKernelGGL(tri_backtransform_quad_kernel<4>, dim3(1), dim3(256), 0, st, t, deg, (const double*)z, evec);
  else if (bt_wave && S <= 256)
    hipLaunchKernelGGL(tri_backtransform_wave_kernel<4>, dim3(1), dim3(64), 0, st, t, deg, (const double*)z, evec);
  else if (bt_wave && S <= 512)
    hipLaunchKernelGGL(tri_backtransform_wave_kernel<8>, dim3(1), dim3(64), 0, st, t, deg, (const double*)z, evec);
  else if (bt_wave && S <= 704)
    hipLaunchKernelGGL(tri_backtransform_wave_kernel<11>, dim3(1), dim3(64), 0, st, t, deg, (const double*)z, evec);
  else if (bt_wave && S <= 1024)
    hipLaunchKernelGGL(tri_backtransform_wave_kernel<16>, dim3(1), dim3(64), 0, st, t, deg, (const double*)z, evec);
  else
    hipLaunchKernelGGL(tri_backtransform_kernel, dim3(1), dim3(1024), 0, st, t, deg, z, evec);
  USC_CHECK_LAUNCH("usc_ncut_fiedler");
  return USC_OK;
}

}  // extern "C"
