// rows.hip — HBM-bound row kernels around the sparse convs (SURVEY.md §8a rows
// B, P, Q3, D1): per-channel statistics for MinkowskiBatchNorm, fused
// BN-apply(+residual)(+ReLU), BN backward, ReLU, 2x2x2 average pooling over a
// child table, row gather and segment mean (torch_scatter.scatter_mean) with a
// stable counting-sort CSR so that every floating-point sum has a fixed order.
// All kernels move 16 B per lane where the channel count allows it.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "scan.h"

namespace usc {

void launch_group_reduce(const float* partial, int G, int64_t numel4, int cout, const float* bias, int accumulate,
                         float* out, hipStream_t st);   // spconv.hip

// ---------------------------------------------------------------------------
// column statistics: two sums per channel over N rows, f64 accumulation,
// block partials -> ordered final reduction (no float atomics).
enum StatMode { STAT_XY = 0, STAT_BN_BWD = 1 };

struct StatArgs {
  const float* x;       // [n,c]
  const float* y;       // STAT_XY: second factor or NULL (x*x); STAT_BN_BWD: dy
  const float* y_out;   // STAT_BN_BWD: forward output for the ReLU mask, or NULL
  const float* mean;    // STAT_BN_BWD
  const float* invstd;  // STAT_BN_BWD
  int64_t n;
  int c;
};

constexpr int kStatMaxBlocks = 1024;
constexpr int kStatUnroll = 4;   // row loads in flight per thread

template <int VEC, int MODE>
__global__ __launch_bounds__(256) void colstats_kernel(StatArgs a, double* __restrict__ partial) {
  extern __shared__ double sh[];  // [RP][c][2]
  const int c = a.c;
  const int CT = c / VEC;        // threads per row
  const int RP = 256 / CT;       // rows per pass
  const int rl = threadIdx.x / CT, cg = threadIdx.x - rl * CT;
  const bool active = rl < RP;
  double s1[VEC], s2[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) s1[v] = s2[v] = 0.0;
  float mu[VEC], is[VEC];
  if (MODE == STAT_BN_BWD && active) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      mu[v] = a.mean[cg * VEC + v];
      is[v] = a.invstd[cg * VEC + v];
    }
  }
  if (active) {
    const int64_t stride = (int64_t)gridDim.x * RP;
    for (int64_t r0 = (int64_t)blockIdx.x * RP + rl; r0 < a.n; r0 += stride * kStatUnroll) {
      // kStatUnroll rows in flight per thread: the loads are issued together (one row per thread and iteration left
      // ~4 MB in flight chip-wide, 2.9 TB/s); the sums keep the row order, so the result is unchanged
      float xv[kStatUnroll][VEC], yv[kStatUnroll][VEC], ov[kStatUnroll][VEC];
#pragma unroll
      for (int u = 0; u < kStatUnroll; ++u) {
        const int64_t r = r0 + u * stride;
        if (r < a.n) {
          const int64_t off = r * c + cg * VEC;
          if (VEC == 4) {
            float4 t = *reinterpret_cast<const float4*>(a.x + off);
            xv[u][0] = t.x; xv[u][1] = t.y; xv[u][2] = t.z; xv[u][3] = t.w;
            if (a.y) {
              float4 w = *reinterpret_cast<const float4*>(a.y + off);
              yv[u][0] = w.x; yv[u][1] = w.y; yv[u][2] = w.z; yv[u][3] = w.w;
            }
            if (MODE == STAT_BN_BWD && a.y_out) {
              float4 w = *reinterpret_cast<const float4*>(a.y_out + off);
              ov[u][0] = w.x; ov[u][1] = w.y; ov[u][2] = w.z; ov[u][3] = w.w;
            }
          } else {
            xv[u][0] = a.x[off];
            if (a.y) yv[u][0] = a.y[off];
            if (MODE == STAT_BN_BWD && a.y_out) ov[u][0] = a.y_out[off];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kStatUnroll; ++u) {
        if (r0 + u * stride < a.n) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            if (MODE == STAT_XY) {
              s1[v] += (double)xv[u][v];
              s2[v] += (double)xv[u][v] * (double)(a.y ? yv[u][v] : xv[u][v]);
            } else {
              float g = yv[u][v];
              if (a.y_out && !(ov[u][v] > 0.f)) g = 0.f;
              const float xhat = (xv[u][v] - mu[v]) * is[v];
              s1[v] += (double)g;
              s2[v] += (double)g * (double)xhat;
            }
          }
        }
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      sh[((int64_t)rl * c + cg * VEC + v) * 2 + 0] = s1[v];
      sh[((int64_t)rl * c + cg * VEC + v) * 2 + 1] = s2[v];
    }
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    double t1 = 0.0, t2 = 0.0;
    for (int r = 0; r < RP; ++r) {
      t1 += sh[((int64_t)r * c + ch) * 2 + 0];
      t2 += sh[((int64_t)r * c + ch) * 2 + 1];
    }
    partial[((int64_t)blockIdx.x * 2 + 0) * c + ch] = t1;
    partial[((int64_t)blockIdx.x * 2 + 1) * c + ch] = t2;
  }
}

// one wave per channel: lanes stride the block partials, fixed shuffle tree -> deterministic
__device__ inline void reduce_partials(const double* __restrict__ partial, int nblocks, int c, int ch, double& t1,
                                       double& t2) {
  const int lane = threadIdx.x & 63;
  double a1 = 0.0, a2 = 0.0;
  for (int b = lane; b < nblocks; b += 64) {
    a1 += partial[((int64_t)b * 2 + 0) * c + ch];
    a2 += partial[((int64_t)b * 2 + 1) * c + ch];
  }
  t1 = wave_reduce_addd(a1);
  t2 = wave_reduce_addd(a2);
}

__global__ __launch_bounds__(64) void colstats_final_kernel(const double* __restrict__ partial, int nblocks, int c,
                                                           double* __restrict__ sum1, double* __restrict__ sum2) {
  const int ch = blockIdx.x;
  double t1, t2;
  reduce_partials(partial, nblocks, c, ch, t1, t2);
  if ((threadIdx.x & 63) == 0) {
    sum1[ch] = t1;
    sum2[ch] = t2;
  }
}

// BatchNorm1d training statistics -> everything the apply/backward kernels need, plus the
// running-stat update (momentum rule of torch.nn.BatchNorm1d: unbiased variance).
struct BnFwdOut {
  const float* gamma; const float* beta;
  float* running_mean; float* running_var;
  float* mean; float* invstd; float* scale; float* shift;
  float eps, momentum;
  int64_t n;
  int64_t* num_batches_tracked;   // incremented once per call when not NULL (nn.BatchNorm's counter)
};
__device__ inline void bn_finalize_channel(int ch, double s1, double s2, const BnFwdOut& o) {
  const double n = (double)o.n;
  const double m = s1 / n;
  double var = s2 / n - m * m;
  if (var < 0.0) var = 0.0;
  const float mean = (float)m;
  const float invstd = (float)(1.0 / sqrt(var + (double)o.eps));
  const float sc = o.gamma[ch] * invstd;
  o.mean[ch] = mean;
  o.invstd[ch] = invstd;
  o.scale[ch] = sc;
  o.shift[ch] = o.beta[ch] - mean * sc;
  if (o.running_mean) {
    const double unbiased = var * (n / (n > 1.0 ? n - 1.0 : 1.0));
    o.running_mean[ch] = (1.f - o.momentum) * o.running_mean[ch] + o.momentum * mean;
    o.running_var[ch] = (1.f - o.momentum) * o.running_var[ch] + o.momentum * (float)unbiased;
  }
  if (ch == 0 && o.num_batches_tracked) *o.num_batches_tracked += 1;
}
__global__ __launch_bounds__(64) void bn_fwd_finalize_kernel(const double* __restrict__ partial, int nblocks, int c,
                                                            BnFwdOut o) {
  const int ch = blockIdx.x;
  double s1, s2;
  reduce_partials(partial, nblocks, c, ch, s1, s2);
  if ((threadIdx.x & 63) == 0) bn_finalize_channel(ch, s1, s2, o);
}
struct BnBwdOut {
  float* dgamma; float* dbeta; float* mean_g; float* mean_gx;
  int64_t n;
  int training, accumulate;
};
__device__ inline void bn_finalize_channel(int ch, double sg, double sgx, const BnBwdOut& o) {
  o.dbeta[ch] = o.accumulate ? o.dbeta[ch] + (float)sg : (float)sg;    // accumulate: into the parameters' .grad buffers
  o.dgamma[ch] = o.accumulate ? o.dgamma[ch] + (float)sgx : (float)sgx;
  o.mean_g[ch] = o.training ? (float)(sg / (double)o.n) : 0.f;
  o.mean_gx[ch] = o.training ? (float)(sgx / (double)o.n) : 0.f;
}
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const double* __restrict__ partial, int nblocks, int c,
                                                            BnBwdOut o) {
  const int ch = blockIdx.x;
  double sg, sgx;
  reduce_partials(partial, nblocks, c, ch, sg, sgx);
  if ((threadIdx.x & 63) == 0) bn_finalize_channel(ch, sg, sgx, o);
}

// Few-row maps (the U-Net's three coarsest levels: 507 ... 2 222 rows at 150k voxels): sums and finalisation in ONE
// launch.  A block owns four channels (one float4 column); its 256 threads stride the rows, then wave 0 folds the
// 256 x 8 partials in a fixed order (thread-major blocks of 8, then a 3-step shuffle tree) and finalises.  The two
// launches this replaces cost 9.4 us at 507 rows against ~5 us for the one; the statistics are the same f64 sums.
// measured (profiles/r03_hbm_bound_kernels.txt): statistics 9.4 -> 4.2 us at 507 rows, 9.8 -> 8.0 at 2 222, slower from
// 9 402; the backward reduce (three inputs) 11.6 -> 5.3 us at 507 rows but 11.9 -> 13.4 at 2 222
static int64_t bn_small_rows(bool backward) {
  static const int64_t knob = [] {
    const char* e = getenv("USC3D_BN_SMALL_ROWS");   // measurement knob; 0 = always the two-launch form
    return e ? (int64_t)atoll(e) : (int64_t)-1;
  }();
  return knob >= 0 ? knob : (backward ? 1024 : 4096);
}
template <int MODE, typename OUT>
__global__ __launch_bounds__(256) void bn_small_kernel(StatArgs a, OUT o) {
  __shared__ double sh[8][256 + 2];
  // workgroups go round-robin over the 8 XCDs (each with its own L2): give XCD x the CONTIGUOUS column groups
  // [x*G/8, (x+1)*G/8) so that the 128-byte lines it pulls are used whole instead of 16 bytes at a time by 8 XCDs
  const int G = gridDim.x, b = blockIdx.x;
  const int grp = (G % 8 == 0) ? (b % 8) * (G / 8) + b / 8 : b;
  const int c = a.c, ch0 = grp * 4, tid = threadIdx.x;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  float mu[4], is[4];
  if (MODE == STAT_BN_BWD) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      mu[v] = a.mean[ch0 + v];
      is[v] = a.invstd[ch0 + v];
    }
  }
  for (int64_t r0 = tid; r0 < a.n; r0 += 256 * kStatUnroll) {
    float4 xv[kStatUnroll], yv[kStatUnroll], ov[kStatUnroll];
#pragma unroll
    for (int u = 0; u < kStatUnroll; ++u) {
      const int64_t r = r0 + u * 256;
      if (r < a.n) {
        const int64_t off = r * c + ch0;
        xv[u] = *reinterpret_cast<const float4*>(a.x + off);
        if (MODE == STAT_BN_BWD) {
          yv[u] = *reinterpret_cast<const float4*>(a.y + off);
          if (a.y_out) ov[u] = *reinterpret_cast<const float4*>(a.y_out + off);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kStatUnroll; ++u) {
      if (r0 + u * 256 < a.n) {
        const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
        if (MODE == STAT_XY) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            s1[v] += (double)xs[v];
            s2[v] += (double)xs[v] * (double)xs[v];
          }
        } else {
          const float gs[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
          const float os[4] = {ov[u].x, ov[u].y, ov[u].z, ov[u].w};
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            float g = gs[v];
            if (a.y_out && !(os[v] > 0.f)) g = 0.f;
            const float xhat = (xs[v] - mu[v]) * is[v];
            s1[v] += (double)g;
            s2[v] += (double)g * (double)xhat;
          }
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    sh[v][tid] = s1[v];
    sh[4 + v][tid] = s2[v];
  }
  __syncthreads();
  if (tid < 64) {
    const int k = tid >> 3, p = tid & 7;      // value k (s1[0..3], s2[0..3]), eighth p of the 256 partials
    double t = 0.0;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) t += sh[k][j * 8 + p];
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
    t += __shfl_xor(t, 4, 64);
    const double second = __shfl(t, (tid + 32) & 63, 64);   // lane 8v: s1[v]; lane 8v+32: s2[v]
    if (tid < 32 && p == 0) {
      bn_finalize_channel(ch0 + k, t, second, o);
    }
  }
}

// eval-mode batch norm: the same four vectors from the running statistics
__global__ void bn_eval_stats_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ rmean, const float* __restrict__ rvar, float eps, int c,
                                     float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                     float* __restrict__ shift) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const float is = 1.f / sqrtf(rvar[ch] + eps);      // torch.rsqrt(running_var + eps) up to rounding
  const float sc = gamma[ch] * is;
  mean[ch] = rmean[ch];
  invstd[ch] = is;
  scale[ch] = sc;
  shift[ch] = beta[ch] - rmean[ch] * sc;
}

static int colstats_blocks(int64_t n, int c, int vec) {
  const int CT = c / vec, RP = 256 / CT;
  int64_t b = ceil_div(n, (int64_t)RP * 4 * kStatUnroll);
  if (b > kStatMaxBlocks) b = kStatMaxBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

template <int MODE>
static int launch_colstats_partials(const StatArgs& a, void* ws, int64_t ws_bytes, hipStream_t st, const char* name,
                                    int* nblocks_out) {
  const int c = a.c;
  USC_REQUIRE(c >= 1 && c <= 1024, "%s: unsupported channel count %d", name, c);
  const int vec = (c % 4 == 0 && c / 4 <= 256) ? 4 : 1;
  USC_REQUIRE(c / vec <= 256, "%s: unsupported channel count %d", name, c);
  const int nb = colstats_blocks(a.n, c, vec);
  USC_REQUIRE(ws_bytes >= (int64_t)nb * 2 * c * 8, "%s: workspace too small", name);
  const int RP = 256 / (c / vec);
  const size_t lds = (size_t)RP * c * 2 * sizeof(double);
  double* partial = (double*)ws;
  if (vec == 4)
    hipLaunchKernelGGL((colstats_kernel<4, MODE>), dim3(nb), dim3(256), lds, st, a, partial);
  else
    hipLaunchKernelGGL((colstats_kernel<1, MODE>), dim3(nb), dim3(256), lds, st, a, partial);
  *nblocks_out = nb;
  return USC_OK;
}

// ---------------------------------------------------------------------------
// The decoder's key sampling (reference models/mask3d.py:306-346), per pass: rows idx[] of the level's feature table,
// of its thresholded attention-mask table and of its positional encodings -> the [B, K, .] inputs of the pass; then
// "a query whose sampled keys are ALL masked attends to everything" (:346) and "padding keys are masked" (:343).
// Two launches instead of three gathers, a sum, a compare, an indexed store and an OR.
//   sample_keys_kernel   grid (ceil(K/16), B), 4 rows per wave, all loads of the 4 rows issued before the stores:
//                        copies the three rows, ANDs the mask bytes of the block's rows -> part[b][blk][q]; padding
//                        rows (k >= n_valid[b]; they repeat a real row) enter the AND with their gathered bits and
//                        are stored as all-masked
//   sample_fix_kernel    same grid: every workgroup ANDs the scene's partials (words, 8 groups of lanes) and clears
//                        the all-masked columns in its own real rows — at random init several queries are masked
//                        everywhere on the fine levels, so the clearing must not be left to one workgroup
constexpr int kSampleRows = 16;       // rows per workgroup (4 waves x 4 rows)
constexpr int kSampleMaxQ = 128;
constexpr int kSampleMaxScenes = 16;
struct SampleArgs {
  const float* feats; const unsigned char* mask; const float* pos; const int64_t* idx;
  float* out_feats; unsigned char* out_mask; float* out_pos; unsigned char* part;
  int c, q, p, K, nblk, qs;           // qs: q rounded up to a multiple of 4 (row stride of part)
  int n_valid[kSampleMaxScenes];
};

__global__ __launch_bounds__(256) void sample_keys_kernel(SampleArgs a) {
  __shared__ unsigned char acc[4][kSampleMaxQ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, k0 = blockIdx.x * kSampleRows + wave * 4;
  const int c4 = a.c >> 2, p4 = a.p >> 2;
  const int nvalid = a.n_valid[b];
  const bool has_pos = a.pos != nullptr;
  // straight-line code (rows past K are clamped to the last row and only their stores are skipped): the 4 rows'
  // loads are all in flight before the first store
  int64_t src[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = k0 + r < a.K ? k0 + r : a.K - 1;
    src[r] = a.idx[(int64_t)b * a.K + k];
  }
  float4 f0[4], f1[4], q0[4], q1[4];
  unsigned char ma[4], mb[4];
  const bool fa = lane < c4, fb = lane + 64 < c4, pa = has_pos && lane < p4, pb = has_pos && lane + 64 < p4;
  const bool qa = lane < a.q, qb = lane + 64 < a.q;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float4* f = reinterpret_cast<const float4*>(a.feats + src[r] * a.c);
    const float4* ps = reinterpret_cast<const float4*>(a.pos + src[r] * a.p);
    const unsigned char* ms = a.mask + src[r] * a.q;       // (q == 0: never dereferenced, qa / qb are false)
    f0[r] = fa ? f[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    f1[r] = fb ? f[lane + 64] : make_float4(0.f, 0.f, 0.f, 0.f);
    q0[r] = pa ? ps[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    q1[r] = pb ? ps[lane + 64] : make_float4(0.f, 0.f, 0.f, 0.f);
    ma[r] = qa ? (ms[lane] ? 1 : 0) : 1;
    mb[r] = qb ? (ms[lane + 64] ? 1 : 0) : 1;
  }
  unsigned char m0 = 1, m1 = 1;       // AND of this wave's rows, bytes lane and lane + 64
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool live = k0 + r < a.K;
    const int64_t o = (int64_t)b * a.K + (live ? k0 + r : 0);
    float4* fo = reinterpret_cast<float4*>(a.out_feats + o * a.c);
    float4* po = reinterpret_cast<float4*>(a.out_pos + o * a.p);
    unsigned char* mo = a.out_mask + o * a.q;
    const bool pad = k0 + r >= nvalid;
    if (live && fa) fo[lane] = f0[r];
    if (live && fb) fo[lane + 64] = f1[r];
    if (live && pa) po[lane] = q0[r];
    if (live && pb) po[lane + 64] = q1[r];
    if (live && qa) mo[lane] = pad ? 1 : ma[r];
    if (live && qb) mo[lane + 64] = pad ? 1 : mb[r];
    m0 &= live ? ma[r] : (unsigned char)1;
    m1 &= live ? mb[r] : (unsigned char)1;
  }
  acc[wave][lane] = m0;
  acc[wave][lane + 64] = m1;
  __syncthreads();
  if (threadIdx.x < a.qs) {
    const int t = threadIdx.x;
    a.part[((int64_t)b * a.nblk + blockIdx.x) * a.qs + t] = acc[0][t] & acc[1][t] & acc[2][t] & acc[3][t];
  }
}

__global__ __launch_bounds__(256) void sample_fix_kernel(SampleArgs a) {
  __shared__ unsigned int red[8][kSampleMaxQ / 4];
  __shared__ unsigned char all_masked[kSampleMaxQ];
  const int b = blockIdx.y, t = threadIdx.x;
  const int w = t & 31, g = t >> 5, qw = a.qs >> 2;
  unsigned int v = 0xFFFFFFFFu;
  if (w < qw) {
    const unsigned int* pp = reinterpret_cast<const unsigned int*>(a.part + (int64_t)b * a.nblk * a.qs) + w;
    for (int j = g; j < a.nblk; j += 8) v &= pp[(int64_t)j * qw];
  }
  red[g][w] = v;
  __syncthreads();
  int any = 0;
  if (t < qw) {
    unsigned int r = red[0][t];
#pragma unroll
    for (int j = 1; j < 8; ++j) r &= red[j][t];
    r &= 0x01010101u;
    reinterpret_cast<unsigned int*>(all_masked)[t] = r;
    // bytes past q in the last word come from the (all-ones) padding of part: ignore them
    const int nb = a.q - 4 * t < 4 ? a.q - 4 * t : 4;
    any = (r & (nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u))) != 0;
  }
  if (!__syncthreads_or(any)) return;          // the usual case later in training: every query sees a key
  const int k0 = blockIdx.x * kSampleRows;
  const int nv = a.n_valid[b] < a.K ? a.n_valid[b] : a.K;
  int rows = nv - k0;
  if (rows > kSampleRows) rows = kSampleRows;
  unsigned char* m = a.out_mask + ((int64_t)b * a.K + k0) * a.q;
  for (int e = t; e < rows * a.q; e += 256)
    if (all_masked[e % a.q]) m[e] = 0;
}

// ---------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                      const float* __restrict__ shift,
                                                      const float* __restrict__ res, int relu, float* __restrict__ y,
                                                      int64_t nvec, int c) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nvec; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = j * VEC;
    const int ch = (int)(e % c);
    if (VEC == 4) {
      float4 v = *reinterpret_cast<const float4*>(x + e);
      const float4 sc = *reinterpret_cast<const float4*>(scale + ch);
      const float4 sf = *reinterpret_cast<const float4*>(shift + ch);
      v.x = v.x * sc.x + sf.x; v.y = v.y * sc.y + sf.y; v.z = v.z * sc.z + sf.z; v.w = v.w * sc.w + sf.w;
      if (res) {
        const float4 r = *reinterpret_cast<const float4*>(res + e);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *reinterpret_cast<float4*>(y + e) = v;
    } else {
      float v = x[e] * scale[ch] + shift[ch];
      if (res) v += res[e];
      if (relu) v = fmaxf(v, 0.f);
      y[e] = v;
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       const float* __restrict__ y_out,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ mean_g,
                                                       const float* __restrict__ mean_gx, float* __restrict__ dx,
                                                       float* __restrict__ dres, int64_t nvec, int c) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nvec; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = j * VEC;
    const int ch0 = (int)(e % c);
    float xv[VEC], gv[VEC], ov[VEC], ox[VEC];
    if (VEC == 4) {
      float4 t = *reinterpret_cast<const float4*>(x + e);
      xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
      t = *reinterpret_cast<const float4*>(dy + e);
      gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w;
      if (y_out) {
        t = *reinterpret_cast<const float4*>(y_out + e);
        ov[0] = t.x; ov[1] = t.y; ov[2] = t.z; ov[3] = t.w;
      }
    } else {
      xv[0] = x[e];
      gv[0] = dy[e];
      if (y_out) ov[0] = y_out[e];
    }
    // the five per-channel vectors as 16-byte loads (one dword load each per element made the kernel LSU-bound:
    // 3.1 TB/s against 6 TB/s for the forward apply kernel with the same traffic pattern)
    float mu[VEC], is[VEC], ga[VEC], mg[VEC], mx[VEC];
    if (VEC == 4) {
      float4 t = *reinterpret_cast<const float4*>(mean + ch0);
      mu[0] = t.x; mu[1] = t.y; mu[2] = t.z; mu[3] = t.w;
      t = *reinterpret_cast<const float4*>(invstd + ch0);
      is[0] = t.x; is[1] = t.y; is[2] = t.z; is[3] = t.w;
      ga[0] = gamma[ch0]; ga[1] = gamma[ch0 + 1]; ga[2] = gamma[ch0 + 2]; ga[3] = gamma[ch0 + 3];   // a parameter: 4-byte aligned only
      t = *reinterpret_cast<const float4*>(mean_g + ch0);
      mg[0] = t.x; mg[1] = t.y; mg[2] = t.z; mg[3] = t.w;
      t = *reinterpret_cast<const float4*>(mean_gx + ch0);
      mx[0] = t.x; mx[1] = t.y; mx[2] = t.z; mx[3] = t.w;
    } else {
      mu[0] = mean[ch0]; is[0] = invstd[ch0]; ga[0] = gamma[ch0]; mg[0] = mean_g[ch0]; mx[0] = mean_gx[ch0];
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float g = gv[v];
      if (y_out && !(ov[v] > 0.f)) g = 0.f;
      gv[v] = g;
      const float xhat = (xv[v] - mu[v]) * is[v];
      ox[v] = ga[v] * is[v] * (g - mg[v] - xhat * mx[v]);
    }
    if (VEC == 4) {
      *reinterpret_cast<float4*>(dx + e) = make_float4(ox[0], ox[1], ox[2], ox[3]);
      if (dres) *reinterpret_cast<float4*>(dres + e) = make_float4(gv[0], gv[1], gv[2], gv[3]);
    } else {
      dx[e] = ox[0];
      if (dres) dres[e] = gv[0];
    }
  }
}

__global__ __launch_bounds__(256) void relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n4; j += (int64_t)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(x)[j];
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    reinterpret_cast<float4*>(y)[j] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    y[e] = fmaxf(x[e], 0.f);
  }
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                      float* __restrict__ dx, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n4; j += (int64_t)gridDim.x * blockDim.x) {
    const float4 o = reinterpret_cast<const float4*>(y)[j];
    float4 g = reinterpret_cast<const float4*>(dy)[j];
    g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
    reinterpret_cast<float4*>(dx)[j] = g;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    dx[e] = y[e] > 0.f ? dy[e] : 0.f;
  }
}

// ---------------------------------------------------------------------------
// row_of (optional): child `ch` reads row row_of[ch] of `in` (leading dimension ld) — the mask module's per-voxel
// logits are rows of the [segments, Q] table, never materialised per voxel.  mask_out (optional): instead of the
// means, sigmoid(mean) < 0.5 as bytes (the attention mask of models/mask3d.py:436).
template <int VEC>
__global__ __launch_bounds__(256) void avgpool_down2_kernel(const float* __restrict__ in, int c, int ld,
                                                           const int64_t* __restrict__ row_of,
                                                           const int32_t* __restrict__ nbr2, int64_t n_coarse,
                                                           float* __restrict__ out, uint8_t* __restrict__ mask_out) {
  const int CT = c / VEC;
  const int64_t total = n_coarse * CT;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = j / CT;
    const int cg = (int)(j - p * CT);
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ch = nbr2[(int64_t)k * n_coarse + p];
      if (ch >= 0) {
        ++cnt;
        const int64_t row = row_of ? row_of[ch] : (int64_t)ch;
        if (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(in + row * ld + cg * 4);
          acc[0] += t.x; acc[1] += t.y; acc[2] += t.z; acc[3] += t.w;
        } else {
          acc[0] += in[row * ld + cg];
        }
      }
    }
    const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
    if (mask_out) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) mask_out[p * c + cg * VEC + v] = (1.f / (1.f + expf(-(acc[v] * inv)))) < 0.5f;
    } else if (VEC == 4) {
      *reinterpret_cast<float4*>(out + p * c + cg * 4) =
          make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
    } else {
      out[p * c + cg] = acc[0] * inv;
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int c,
                                                         const int64_t* __restrict__ idx, int64_t n,
                                                         float* __restrict__ out) {
  const int CT = c / VEC;
  const int64_t total = n * CT;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = j / CT;
    const int cg = (int)(j - r * CT);
    const int64_t sr = idx[r];
    if (VEC == 4)
      *reinterpret_cast<float4*>(out + r * c + cg * 4) = *reinterpret_cast<const float4*>(src + sr * c + cg * 4);
    else
      out[r * c + cg] = src[sr * c + cg];
  }
}

// ---------------------------------------------------------------------------
// Stable counting sort of rows by segment id -> CSR (order, seg_off).
constexpr int kSegTile = 1024;

__global__ __launch_bounds__(256) void seg_hist_kernel(const int64_t* __restrict__ seg, int64_t n, int64_t S,
                                                      int32_t* __restrict__ hist /*[T][S]*/) {
  const int64_t t = blockIdx.x;
  const int64_t b = t * kSegTile;
  for (int j = threadIdx.x; j < kSegTile; j += 256) {
    const int64_t i = b + j;
    if (i < n) atomicAdd(&hist[t * S + seg[i]], 1);
  }
}
// per segment: exclusive scan over tiles (in place), total -> counts[s]
__global__ __launch_bounds__(256) void seg_tilescan_kernel(int32_t* __restrict__ hist, int64_t T, int64_t S,
                                                          int32_t* __restrict__ counts) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  int run = 0;
  for (int64_t t = 0; t < T; ++t) {
    const int v = hist[t * S + s];
    hist[t * S + s] = run;
    run += v;
  }
  counts[s] = run;
}
struct CountVal {
  const int32_t* counts;
  __device__ int operator()(int64_t i) const { return counts[i]; }
};
struct SegOffEmit {
  int64_t* seg_off;
  __device__ void operator()(int64_t i, int64_t pos, int) const { seg_off[i] = pos; }
};
// One wave per tile walks its rows in order.  Inside a 64-row chunk the rank of a row among the
// lower lanes holding the same id (and the id's chunk total) comes from a fixed 64-step shuffle
// sweep — no data-dependent loop — so the placement is stable and its cost independent of how many
// distinct ids a chunk holds.
__global__ __launch_bounds__(64) void seg_place_kernel(const int64_t* __restrict__ seg, int64_t n, int64_t S,
                                                      int32_t* __restrict__ base /*[T][S] tile offsets*/,
                                                      const int64_t* __restrict__ seg_off,
                                                      int64_t* __restrict__ order) {
  const int64_t t = blockIdx.x;
  const int lane = threadIdx.x;
  volatile int32_t* mybase = base + t * S;   // re-read after this wave's own stores
  for (int ch = 0; ch < kSegTile / 64; ++ch) {
    const int64_t i = t * kSegTile + ch * 64 + lane;
    const bool act = i < n;
    const int sid = act ? (int)seg[i] : -1 - lane;   // inactive lanes get unique negative ids
    int rank = 0, total = 0;
#pragma unroll 16
    for (int j = 0; j < 64; ++j) {
      const int other = __shfl(sid, j, 64);
      const int same = other == sid;
      total += same;
      rank += same & (j < lane);
    }
    if (act) {
      const int old = mybase[sid];                  // all lanes of one id read the same word ...
      order[seg_off[sid] + old + rank] = i;
      if (rank == total - 1) mybase[sid] = old + total;   // ... and only its last lane advances it
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// dst[idx[i], :] += src[i, :]   (float atomics; used for the backward of row gathers whose
// indices are a sampled subset — duplicates only occur on masked padding rows)
template <int VEC>
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, int c,
                                                              const int64_t* __restrict__ idx, int64_t n,
                                                              float* __restrict__ dst) {
  const int CT = c / VEC;
  const int64_t total = n * CT;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = j / CT;
    const int cg = (int)(j - r * CT);
    const int64_t d = idx[r];
#pragma unroll
    for (int v = 0; v < VEC; ++v) unsafeAtomicAdd(dst + d * c + cg * VEC + v, src[r * c + cg * VEC + v]);   // the hardware's
    // global_atomic_add_f32 (plain atomicAdd on float compiles to a compare-and-swap loop: 0.9 TB/s on a permutation)
  }
}

// dst[idx[i], :] = src[i, :] for an index set WITHOUT duplicates (a permutation, a sampled subset): plain 16-byte
// stores — the atomic form above is bound by the L2's atomic units (0.9 TB/s of algorithmic bytes at 148 k rows)
template <int VEC>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, int c,
                                                          const int64_t* __restrict__ idx, int64_t n,
                                                          float* __restrict__ dst) {
  const int CT = c / VEC;
  const int64_t total = n * CT;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = j / CT;
    const int cg = (int)(j - r * CT);
    const int64_t d = idx[r];
    if (VEC == 4)
      *reinterpret_cast<float4*>(dst + d * c + cg * 4) = *reinterpret_cast<const float4*>(src + r * c + cg * 4);
    else
      dst[d * c + cg] = src[r * c + cg];
  }
}

// dst[idx[i], :] += src[i, :], duplicate-free index set: plain read-modify-write (launches on one stream are ordered, so
// several sampled subsets of the same table may be accumulated one launch after the other without atomics)
template <int VEC>
__global__ __launch_bounds__(256) void scatter_rows_rmw_kernel(const float* __restrict__ src, int c,
                                                              const int64_t* __restrict__ idx, int64_t n,
                                                              float* __restrict__ dst) {
  const int CT = c / VEC;
  const int64_t total = n * CT;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = j / CT;
    const int cg = (int)(j - r * CT);
    const int64_t d = idx[r];
    if (VEC == 4) {
      float4* o = reinterpret_cast<float4*>(dst + d * c + cg * 4);
      const float4 v = *reinterpret_cast<const float4*>(src + r * c + cg * 4);
      const float4 t = *o;
      *o = make_float4(t.x + v.x, t.y + v.y, t.z + v.z, t.w + v.w);
    } else {
      dst[d * c + cg] += src[r * c + cg];
    }
  }
}

// mode bit 0: use only rows with a non-zero entry; bit 1: per-channel MAX instead of the mean (N1 'max' mode)
__global__ __launch_bounds__(256) void segment_mean_fwd_kernel(const float* __restrict__ src, int c,
                                                              const int64_t* __restrict__ order,
                                                              const int64_t* __restrict__ seg_off, int64_t S,
                                                              int mode, float* __restrict__ out,
                                                              int64_t* __restrict__ nz_cnt) {
  // one wave per segment; lanes stride the channels (up to 8 chunks of 64 held in registers, more in further
  // sweeps); rows visited once per sweep in CSR order
  const int lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= S) return;
  const bool only_nonzero = mode & 1, take_max = mode & 2;
  const int64_t b = seg_off[s], e = seg_off[s + 1];
  constexpr int CH = 8;
  for (int base = 0; base < c; base += 64 * CH) {
    float acc[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[j] = take_max ? -INFINITY : 0.f;
    int64_t cnt = 0;
    for (int64_t q = b; q < e; ++q) {
      const float* row = src + order[q] * (int64_t)c;
      float v[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int ch = base + 64 * j + lane;
        v[j] = ch < c ? row[ch] : 0.f;
      }
      bool use = true;
      if (only_nonzero) {
        // valid row = torch.any(features != 0, dim=-1)  (unscene3d_pseudo_main.py:362)
        bool nz = false;
        if (c <= 64 * CH) {
#pragma unroll
          for (int j = 0; j < CH; ++j) nz |= v[j] != 0.f;
        } else {
          for (int cc = lane; cc < c; cc += 64) nz |= row[cc] != 0.f;
        }
        use = __any(nz);
      }
      if (use) {
        ++cnt;
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[j] = take_max ? fmaxf(acc[j], v[j]) : acc[j] + v[j];
      }
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int ch = base + 64 * j + lane;
      if (ch < c) out[s * c + ch] = cnt > 0 ? (take_max ? acc[j] : acc[j] / (float)cnt) : 0.f;
    }
    if (base == 0 && lane == 0 && nz_cnt) nz_cnt[s] = cnt;
  }
}

// c % 4 == 0 and c <= 128: a half-wave (32 lanes x 16 B) covers one row; the eight half-waves of a workgroup walk
// interleaved rows of ONE segment, four rows in flight each (a wave per segment with one dependent
// index -> row round trip per iteration ran at 1.3 TB/s on 609 segments of ~244 rows); partial sums are combined in
// a fixed order (the four chains pairwise, then the half-waves ascending), so the mean is bit-reproducible.
__global__ __launch_bounds__(256) void segment_mean_fwd_vec_kernel(const float* __restrict__ src, int c,
                                                                  const int64_t* __restrict__ order,
                                                                  const int64_t* __restrict__ seg_off, int64_t S,
                                                                  float* __restrict__ out) {
  __shared__ float4 red[8][32];
  const int i = threadIdx.x & 31, hw = threadIdx.x >> 5;
  const int64_t s = blockIdx.x;
  const int64_t b = seg_off[s], e = seg_off[s + 1];
  const bool on = 4 * i < c;
  float4 a[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t q = b + hw; q < e; q += 32) {
    int64_t r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = q + 8 * u < e ? order[q + 8 * u] : -1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (on && r[u] >= 0) {
        const float4 v = *reinterpret_cast<const float4*>(src + r[u] * c + 4 * i);
        a[u].x += v.x; a[u].y += v.y; a[u].z += v.z; a[u].w += v.w;
      }
    }
  }
  red[hw][i] = make_float4((a[0].x + a[1].x) + (a[2].x + a[3].x), (a[0].y + a[1].y) + (a[2].y + a[3].y),
                           (a[0].z + a[1].z) + (a[2].z + a[3].z), (a[0].w + a[1].w) + (a[2].w + a[3].w));
  __syncthreads();
  if (hw == 0 && on) {
    float4 t = red[0][i];
#pragma unroll
    for (int w = 1; w < 8; ++w) { t.x += red[w][i].x; t.y += red[w][i].y; t.z += red[w][i].z; t.w += red[w][i].w; }
    const float inv = e > b ? 1.f / (float)(e - b) : 0.f;
    *reinterpret_cast<float4*>(out + s * c + 4 * i) = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void segment_mean_bwd_kernel(const float* __restrict__ dout, int c,
                                                              const int64_t* __restrict__ seg,
                                                              const int64_t* __restrict__ seg_off, int64_t n,
                                                              float* __restrict__ dsrc) {
  const int CT = c / VEC;
  const int64_t total = n * CT;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = j / CT;
    const int cg = (int)(j - r * CT);
    const int64_t s = seg[r];
    const float inv = 1.f / (float)(seg_off[s + 1] - seg_off[s]);
    if (VEC == 4) {
      float4 v = *reinterpret_cast<const float4*>(dout + s * c + cg * 4);
      v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
      *reinterpret_cast<float4*>(dsrc + r * c + cg * 4) = v;
    } else {
      dsrc[r * c + cg] = dout[s * c + cg] * inv;
    }
  }
}

// ---------------------------------------------------------------------------
// AdamW over flat parameter / gradient / moment buffers (trainer/trainer.py:953-966 configures torch.optim.AdamW;
// operation order of PyTorch's fused kernel).  One pass: reads p, g, m, v and writes p, m, v — 28 B per parameter.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n4, int64_t n,
                                                   float lr, float beta1, float beta2, float eps, float wd,
                                                   float step_size, float bc2_sqrt) {
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    pp -= lr * wd * pp;
    mm = mm + (1.0f - beta1) * (gg - mm);      // lerp(m, g, 1 - beta1)
    vv = beta2 * vv + (1.0f - beta2) * gg * gg;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp -= step_size * mm / denom;
  };
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += (int64_t)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[j], mv = reinterpret_cast<float4*>(m)[j], vv = reinterpret_cast<float4*>(v)[j];
    const float4 gv = reinterpret_cast<const float4*>(g)[j];
    upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
    reinterpret_cast<float4*>(p)[j] = pv; reinterpret_cast<float4*>(m)[j] = mv; reinterpret_cast<float4*>(v)[j] = vv;
  }
  if (blockIdx.x == 0) {
    const int64_t j = n4 * 4 + threadIdx.x;
    if (j < n) upd(p[j], g[j], m[j], v[j]);
  }
}


// ---------------------------------------------------------------------------
// TILE FORM of the batch norm around a split-K convolution (maps of a few hundred to ~12 k rows: the U-Net's three
// coarsest levels).  There every launch sits on its ~4.5 us floor, and a conv -> BN -> ReLU unit was FOUR of them after
// the convolution itself had written its G partial slices: slice reduction, statistics, finalisation, apply (three on
// the maps <= 4 096 rows).  Here it is two, neither with a grid-wide wait:
//   bn_tile_stats_kernel   (row tile x 32-column block): y = sum of the G slices in slice order (the order of
//                          group_reduce_kernel: bit-identical y), written once, and the tile's f64 column sums
//                          sum(y), sum(y^2) -> tpart[tile][2][c].  <= 64 row tiles per map.
//   bn_tile_apply_kernel   every workgroup first adds the <= 64 tile sums of all c columns in tile order (the same order
//                          in every workgroup: identical statistics; ~130 independent L2 loads per thread) and finalises
//                          into LDS; workgroup 0 also writes mean / invstd / scale / shift and the running statistics.
//                          Then scale/shift (+ residual) (+ ReLU) over its rows.
// Backward the same pair: bn_tile_bwd_stats_kernel forms dout = (accumulate ? dout : 0) + sum of the input-gradient
// slices the NEXT convolution's backward left behind (usc_program_run hands them over), applies the ReLU mask, writes
// the masked gradient once and the tile sums sum(g), sum(g * xhat); bn_tile_bwd_dx_kernel finalises (workgroup 0 adds
// dgamma / dbeta) and writes dy.
constexpr int kBnTileCols4 = 8;            // float4 columns per workgroup (32 channels, 128 bytes per row)
constexpr int kBnTileRowLanes = 256 / kBnTileCols4;
constexpr int kBnTileMaxTiles = 64;

struct BnTileStats {
  const float* partial;   // [G][n][c] slices, or NULL
  int G;
  int64_t slice;          // floats per slice
  const float* acc_in;    // backward: added first when not NULL (the `accumulate` of the slice reduction)
  const float* src;       // G == 0: the finished tensor (forward: y, backward: dout)
  float* dst;             // forward: y (G > 0); backward: masked gradient (or NULL: nothing to write)
  const float* x;         // backward: the conv output y of the forward pass (for xhat)
  const float* y_out;     // backward: forward output after ReLU (mask) or NULL
  const float* mean; const float* invstd;   // backward
  int64_t n; int c; int tr;
  double* tpart;          // [ntiles][2][c]
};

template <bool BWD>
__global__ __launch_bounds__(256) void bn_tile_stats_kernel(BnTileStats a) {
  __shared__ double sh[2][kBnTileRowLanes][33];
  const int tid = threadIdx.x, rl = tid / kBnTileCols4, cq = tid % kBnTileCols4;
  const int tile = blockIdx.x, cb = blockIdx.y;
  const int col = (cb * kBnTileCols4 + cq) * 4;
  const int64_t r_begin = (int64_t)tile * a.tr;
  int64_t r_end = r_begin + a.tr;
  if (r_end > a.n) r_end = a.n;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  float mu[4] = {0, 0, 0, 0}, is[4] = {0, 0, 0, 0};
  if (BWD) {
    const float4 m4 = *reinterpret_cast<const float4*>(a.mean + col);
    const float4 i4 = *reinterpret_cast<const float4*>(a.invstd + col);
    mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w;
    is[0] = i4.x; is[1] = i4.y; is[2] = i4.z; is[3] = i4.w;
  }
  for (int64_t r = r_begin + rl; r < r_end; r += kBnTileRowLanes) {
    const int64_t off = r * a.c + col;
    float4 v;
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), ov = make_float4(1.f, 1.f, 1.f, 1.f);
    if (BWD) {
      xv = *reinterpret_cast<const float4*>(a.x + off);
      if (a.y_out) ov = *reinterpret_cast<const float4*>(a.y_out + off);
    }
    if (a.G == 0) {
      v = *reinterpret_cast<const float4*>(a.src + off);
    } else {
      v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (BWD && a.acc_in) v = *reinterpret_cast<const float4*>(a.acc_in + off);
      const float* pp = a.partial + off;
      int g = 0;
      for (; g + 8 <= a.G; g += 8) {       // eight slices in flight; the adds keep the order g ascending
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(pp + (int64_t)(g + u) * a.slice);
#pragma unroll
        for (int u = 0; u < 8; ++u) { v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w; }
      }
      for (; g < a.G; ++g) {
        const float4 t = *reinterpret_cast<const float4*>(pp + (int64_t)g * a.slice);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
    }
    float vs[4] = {v.x, v.y, v.z, v.w};
    if (BWD) {
      const float os[4] = {ov.x, ov.y, ov.z, ov.w};
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (a.y_out && !(os[q] > 0.f)) vs[q] = 0.f;
        const float xhat = (xs[q] - mu[q]) * is[q];
        s1[q] += (double)vs[q];
        s2[q] += (double)vs[q] * (double)xhat;
      }
      if (a.dst) *reinterpret_cast<float4*>(a.dst + off) = make_float4(vs[0], vs[1], vs[2], vs[3]);
    } else {
      if (a.G > 0) *reinterpret_cast<float4*>(a.dst + off) = v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s1[q] += (double)vs[q];
        s2[q] += (double)vs[q] * (double)vs[q];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sh[0][rl][cq * 4 + q] = s1[q];
    sh[1][rl][cq * 4 + q] = s2[q];
  }
  __syncthreads();
  if (tid < 64) {
    const int which = tid >> 5, j = tid & 31;
    double t = 0.0;
#pragma unroll 8
    for (int r = 0; r < kBnTileRowLanes; ++r) t += sh[which][r][j];
    a.tpart[((int64_t)tile * 2 + which) * a.c + cb * 32 + j] = t;
  }
}

// sum of the tile partials of one column, tile order; 16 tiles' loads in flight (a plain loop is one L2 round trip per
// tile: ~60 dependent trips for the 9 402-row level)
__device__ inline void tile_sums(const double* __restrict__ tpart, int ntiles, int c, int ch, double& s1, double& s2) {
  s1 = 0.0; s2 = 0.0;
  for (int t0 = 0; t0 < ntiles; t0 += 16) {
    double a1[16], a2[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int t = t0 + u < ntiles ? t0 + u : ntiles - 1;
      a1[u] = tpart[((int64_t)t * 2 + 0) * c + ch];
      a2[u] = tpart[((int64_t)t * 2 + 1) * c + ch];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (t0 + u < ntiles) { s1 += a1[u]; s2 += a2[u]; }
    }
  }
}

struct BnTileApply {
  const double* tpart; int ntiles;
  BnFwdOut o;
  const float* y; const float* res; int relu; float* out; int64_t nvec; int c;
};
__global__ __launch_bounds__(256) void bn_tile_apply_kernel(BnTileApply a) {
  extern __shared__ float tile_sh[];            // scale[c] | shift[c]
  const int c = a.c;
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    double s1, s2;
    tile_sums(a.tpart, a.ntiles, c, ch, s1, s2);
    const double n = (double)a.o.n;
    const double m = s1 / n;
    double var = s2 / n - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)m;
    const float invstd = (float)(1.0 / sqrt(var + (double)a.o.eps));
    const float sc = a.o.gamma[ch] * invstd;
    const float sf = a.o.beta[ch] - mean * sc;
    tile_sh[ch] = sc;
    tile_sh[c + ch] = sf;
    if (blockIdx.x == 0) bn_finalize_channel(ch, s1, s2, a.o);      // the same numbers, written once
  }
  __syncthreads();
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < a.nvec; j += (int64_t)gridDim.x * 256) {
    const int64_t e = j * 4;
    const int ch = (int)(e % c);
    float4 v = *reinterpret_cast<const float4*>(a.y + e);
    const float4 sc = *reinterpret_cast<const float4*>(tile_sh + ch);
    const float4 sf = *reinterpret_cast<const float4*>(tile_sh + c + ch);
    v.x = v.x * sc.x + sf.x; v.y = v.y * sc.y + sf.y; v.z = v.z * sc.z + sf.z; v.w = v.w * sc.w + sf.w;
    if (a.res) {
      const float4 r = *reinterpret_cast<const float4*>(a.res + e);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<float4*>(a.out + e) = v;
  }
}

struct BnTileDx {
  const double* tpart; int ntiles;
  BnBwdOut o;                 // dgamma, dbeta (workgroup 0), mean_g / mean_gx unused (NULL)
  const float* g;             // masked gradient (or dout when mask_src is given)
  const float* mask_src;      // forward output after ReLU when g is still unmasked, else NULL
  const float* x; const float* mean; const float* invstd; const float* gamma;
  float* dx; int64_t nvec; int c;
};
__global__ __launch_bounds__(256) void bn_tile_bwd_dx_kernel(BnTileDx a) {
  extern __shared__ float tile_sh[];            // mean_g[c] | mean_gx[c]
  const int c = a.c;
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    double sg, sgx;
    tile_sums(a.tpart, a.ntiles, c, ch, sg, sgx);
    tile_sh[ch] = a.o.training ? (float)(sg / (double)a.o.n) : 0.f;
    tile_sh[c + ch] = a.o.training ? (float)(sgx / (double)a.o.n) : 0.f;
    if (blockIdx.x == 0) {
      a.o.dbeta[ch] = a.o.accumulate ? a.o.dbeta[ch] + (float)sg : (float)sg;
      a.o.dgamma[ch] = a.o.accumulate ? a.o.dgamma[ch] + (float)sgx : (float)sgx;
    }
  }
  __syncthreads();
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < a.nvec; j += (int64_t)gridDim.x * 256) {
    const int64_t e = j * 4;
    const int ch0 = (int)(e % c);
    const float4 xv = *reinterpret_cast<const float4*>(a.x + e);
    float4 gv = *reinterpret_cast<const float4*>(a.g + e);
    if (a.mask_src) {
      const float4 ov = *reinterpret_cast<const float4*>(a.mask_src + e);
      if (!(ov.x > 0.f)) gv.x = 0.f;
      if (!(ov.y > 0.f)) gv.y = 0.f;
      if (!(ov.z > 0.f)) gv.z = 0.f;
      if (!(ov.w > 0.f)) gv.w = 0.f;
    }
    const float4 mu = *reinterpret_cast<const float4*>(a.mean + ch0);
    const float4 is = *reinterpret_cast<const float4*>(a.invstd + ch0);
    const float4 mg = *reinterpret_cast<const float4*>(tile_sh + ch0);
    const float4 mx = *reinterpret_cast<const float4*>(tile_sh + c + ch0);
    const float ga0 = a.gamma[ch0], ga1 = a.gamma[ch0 + 1], ga2 = a.gamma[ch0 + 2], ga3 = a.gamma[ch0 + 3];
    float4 o;
    o.x = ga0 * is.x * (gv.x - mg.x - (xv.x - mu.x) * is.x * mx.x);
    o.y = ga1 * is.y * (gv.y - mg.y - (xv.y - mu.y) * is.y * mx.y);
    o.z = ga2 * is.z * (gv.z - mg.z - (xv.z - mu.z) * is.z * mx.z);
    o.w = ga3 * is.w * (gv.w - mg.w - (xv.w - mu.w) * is.w * mx.w);
    *reinterpret_cast<float4*>(a.dx + e) = o;
  }
}

struct BnTileGeom { int tr, ntiles, ncb; };
static BnTileGeom bn_tile_geom(int64_t n, int c) {
  BnTileGeom g;
  const int64_t per = ceil_div(n, (int64_t)kBnTileMaxTiles);
  g.tr = (int)(ceil_div(per, (int64_t)kBnTileRowLanes) * kBnTileRowLanes);
  g.ntiles = (int)ceil_div(n, (int64_t)g.tr);
  g.ncb = c / 32;
  return g;
}
static int bn_apply_grid(int64_t nvec) {
  int64_t g = ceil_div(nvec, (int64_t)256 * 2);
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace usc

using namespace usc;

extern "C" {

int64_t usc_colstats_ws_bytes(int64_t n, int32_t c) { (void)n; return (int64_t)kStatMaxBlocks * 2 * c * 8; }

int usc_colstats(const float* x, const float* y, int64_t n, int32_t c, double* sum1, double* sum2, void* ws,
                 int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(x && sum1 && sum2 && ws && n >= 0, "usc_colstats: bad argument");
  StatArgs a{x, y, nullptr, nullptr, nullptr, n, (int)c};
  int nb = 0;
  int rc = launch_colstats_partials<STAT_XY>(a, ws, ws_bytes, as_stream(s), "usc_colstats", &nb);
  if (rc) return rc;
  hipLaunchKernelGGL(colstats_final_kernel, dim3((unsigned)c), dim3(64), 0, as_stream(s), (const double*)ws, nb, (int)c,
                     sum1, sum2);
  USC_CHECK_LAUNCH("usc_colstats");
  return USC_OK;
}

int usc_bn_forward_stats(const float* x, int64_t n, int32_t c, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                         float* mean, float* invstd, float* scale, float* shift, void* ws, int64_t ws_bytes,
                         usc_stream_t s) {
  USC_REQUIRE(x && gamma && beta && mean && invstd && scale && shift && ws && n >= 1, "usc_bn_forward_stats: bad argument");
  USC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "usc_bn_forward_stats: running stats mismatch");
  StatArgs a{x, nullptr, nullptr, nullptr, nullptr, n, (int)c};
  BnFwdOut o{gamma, beta, running_mean, running_var, mean, invstd, scale, shift, eps, momentum, n, num_batches_tracked};
  if (n <= bn_small_rows(false) && c % 4 == 0 && c >= 4) {
    hipLaunchKernelGGL((bn_small_kernel<STAT_XY, BnFwdOut>), dim3((unsigned)(c / 4)), dim3(256), 0, as_stream(s), a, o);
    USC_CHECK_LAUNCH("usc_bn_forward_stats");
    return USC_OK;
  }
  int nb = 0;
  int rc = launch_colstats_partials<STAT_XY>(a, ws, ws_bytes, as_stream(s), "usc_bn_forward_stats", &nb);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((unsigned)c), dim3(64), 0, as_stream(s), (const double*)ws, nb, (int)c, o);
  USC_CHECK_LAUNCH("usc_bn_forward_stats");
  return USC_OK;
}

int usc_bn_eval_stats(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                      int32_t c, float* mean, float* invstd, float* scale, float* shift, usc_stream_t s) {
  USC_REQUIRE(gamma && beta && running_mean && running_var && mean && invstd && scale && shift && c >= 1,
              "usc_bn_eval_stats: bad argument");
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((unsigned)ceil_div(c, 256)), dim3(256), 0, as_stream(s), gamma, beta,
                     running_mean, running_var, eps, (int)c, mean, invstd, scale, shift);
  USC_CHECK_LAUNCH("usc_bn_eval_stats");
  return USC_OK;
}

int usc_bn_backward_reduce(const float* x, const float* dy, const float* y_out, const float* mean, const float* invstd,
                           int64_t n, int32_t c, int32_t training, int32_t accumulate, float* dgamma, float* dbeta,
                           float* mean_g, float* mean_gxhat, void* ws, int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(x && dy && mean && invstd && dgamma && dbeta && mean_g && mean_gxhat && ws && n >= 1,
              "usc_bn_backward_reduce: bad argument");
  StatArgs a{x, dy, y_out, mean, invstd, n, (int)c};
  BnBwdOut o{dgamma, dbeta, mean_g, mean_gxhat, n, (int)training, (int)accumulate};
  if (n <= bn_small_rows(true) && c % 4 == 0 && c >= 4) {
    hipLaunchKernelGGL((bn_small_kernel<STAT_BN_BWD, BnBwdOut>), dim3((unsigned)(c / 4)), dim3(256), 0, as_stream(s), a, o);
    USC_CHECK_LAUNCH("usc_bn_backward_reduce");
    return USC_OK;
  }
  int nb = 0;
  int rc = launch_colstats_partials<STAT_BN_BWD>(a, ws, ws_bytes, as_stream(s), "usc_bn_backward_reduce", &nb);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)c), dim3(64), 0, as_stream(s), (const double*)ws, nb, (int)c, o);
  USC_CHECK_LAUNCH("usc_bn_backward_reduce");
  return USC_OK;
}

int usc_bn_apply(const float* x, const float* scale, const float* shift, const float* residual, int32_t relu, float* y,
                 int64_t n, int32_t c, usc_stream_t s) {
  USC_REQUIRE(n >= 0 && c >= 1, "usc_bn_apply: bad sizes");
  if (n == 0) return USC_OK;
  USC_REQUIRE(x && scale && shift && y, "usc_bn_apply: null pointer");
  const int64_t numel = n * c;
  if (c % 4 == 0)
    hipLaunchKernelGGL((bn_apply_kernel<4>), dim3(stream_grid(numel / 4, 256)), dim3(256), 0, as_stream(s), x, scale,
                       shift, residual, (int)relu, y, numel / 4, (int)c);
  else
    hipLaunchKernelGGL((bn_apply_kernel<1>), dim3(stream_grid(numel, 256)), dim3(256), 0, as_stream(s), x, scale,
                       shift, residual, (int)relu, y, numel, (int)c);
  USC_CHECK_LAUNCH("usc_bn_apply");
  return USC_OK;
}

int usc_bn_backward_dx(const float* x, const float* dy, const float* y_out, const float* mean, const float* invstd,
                       const float* gamma, const float* mean_g, const float* mean_gxhat, float* dx, float* dres,
                       int64_t n, int32_t c, usc_stream_t s) {
  USC_REQUIRE(n >= 0 && c >= 1, "usc_bn_backward_dx: bad sizes");
  if (n == 0) return USC_OK;
  USC_REQUIRE(x && dy && mean && invstd && gamma && mean_g && mean_gxhat && dx, "usc_bn_backward_dx: null pointer");
  const int64_t numel = n * c;
  if (c % 4 == 0)
    hipLaunchKernelGGL((bn_bwd_dx_kernel<4>), dim3(stream_grid(numel / 4, 256)), dim3(256), 0, as_stream(s), x, dy,
                       y_out, mean, invstd, gamma, mean_g, mean_gxhat, dx, dres, numel / 4, (int)c);
  else
    hipLaunchKernelGGL((bn_bwd_dx_kernel<1>), dim3(stream_grid(numel, 256)), dim3(256), 0, as_stream(s), x, dy, y_out,
                       mean, invstd, gamma, mean_g, mean_gxhat, dx, dres, numel, (int)c);
  USC_CHECK_LAUNCH("usc_bn_backward_dx");
  return USC_OK;
}

int64_t usc_bn_tile_max_rows(void) {
  static const int64_t knob = getenv("USC3D_BN_TILE_ROWS") ? (int64_t)atoll(getenv("USC3D_BN_TILE_ROWS")) : (int64_t)4096;
  // 0: the tile form is off (A/B switch).  4 096: the two coarsest levels of a 150 k-voxel scene; measured per unit
  // (tools/bn_tile_bench.py) 17.3 -> 11.5 / 20.9 -> 9.5 us forward / backward at 507 rows, 18.5 -> 15.4 / 24.9 -> 17.9 at
  // 2 222 x 128; at 9 402 rows the old launches are as fast (21.1 vs 22.3 / 24.3 vs 25.9): left as they were
  return knob;
}

// what the kernels cover (any number of rows: <= 64 row tiles of whatever height); usc_bn_tile_max_rows() is the POLICY
// the unit calls apply on top (units.hip)
int usc_bn_tile_ok(int64_t n, int32_t c) { return n >= 1 && c >= 32 && c % 32 == 0 && c <= 1024; }

int64_t usc_bn_tile_ws_bytes(int32_t c) { return (int64_t)kBnTileMaxTiles * 2 * c * 8; }

int usc_group_reduce(const float* partial, int32_t G, int64_t n, int32_t c, const float* bias, int32_t accumulate,
                     float* out, usc_stream_t s) {
  USC_REQUIRE(partial && out && G >= 1 && n >= 0 && c >= 4 && c % 4 == 0, "usc_group_reduce: bad argument");
  if (n == 0) return USC_OK;
  launch_group_reduce(partial, (int)G, n * c / 4, (int)c, bias, (int)accumulate, out, as_stream(s));
  USC_CHECK_LAUNCH("usc_group_reduce");
  return USC_OK;
}

int usc_bn_tile_forward(const float* partial, int32_t G, float* y, int64_t n, int32_t c, const float* gamma,
                        const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                        const float* residual, int32_t relu, float* out, void* ws, int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(usc_bn_tile_ok(n, c), "usc_bn_tile_forward: map of %lld rows x %d channels is outside the tile form", (long long)n, (int)c);
  USC_REQUIRE(y && gamma && beta && mean && invstd && scale && shift && out && ws && G >= 0 && (G == 0 || partial),
              "usc_bn_tile_forward: null pointer");
  USC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "usc_bn_tile_forward: running stats mismatch");
  USC_REQUIRE(ws_bytes >= usc_bn_tile_ws_bytes(c), "usc_bn_tile_forward: workspace too small");
  hipStream_t st = as_stream(s);
  const BnTileGeom g = bn_tile_geom(n, c);
  BnTileStats a{};
  a.partial = partial; a.G = G; a.slice = n * c; a.src = y; a.dst = y; a.n = n; a.c = c; a.tr = g.tr; a.tpart = (double*)ws;
  hipLaunchKernelGGL((bn_tile_stats_kernel<false>), dim3((unsigned)g.ntiles, (unsigned)g.ncb), dim3(256), 0, st, a);
  BnTileApply b{};
  b.tpart = (const double*)ws; b.ntiles = g.ntiles;
  b.o = BnFwdOut{gamma, beta, running_mean, running_var, mean, invstd, scale, shift, eps, momentum, n, num_batches_tracked};
  b.y = y; b.res = residual; b.relu = relu; b.out = out; b.nvec = n * c / 4; b.c = c;
  hipLaunchKernelGGL(bn_tile_apply_kernel, dim3((unsigned)bn_apply_grid(b.nvec)), dim3(256), (size_t)2 * c * 4, st, b);
  USC_CHECK_LAUNCH("usc_bn_tile_forward");
  return USC_OK;
}

int usc_bn_tile_backward(const float* partial, int32_t G, int32_t accumulate, float* dout, const float* y, const float* y_out,
                         const float* mean, const float* invstd, const float* gamma, int64_t n, int32_t c, int32_t training,
                         int32_t dbn_accumulate, float* dgamma, float* dbeta, float* dy, float* dres, void* ws,
                         int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(usc_bn_tile_ok(n, c), "usc_bn_tile_backward: map of %lld rows x %d channels is outside the tile form", (long long)n, (int)c);
  USC_REQUIRE(dout && y && mean && invstd && gamma && dgamma && dbeta && dy && ws && G >= 0 && (G == 0 || partial),
              "usc_bn_tile_backward: null pointer");
  USC_REQUIRE(ws_bytes >= usc_bn_tile_ws_bytes(c), "usc_bn_tile_backward: workspace too small");
  hipStream_t st = as_stream(s);
  const BnTileGeom g = bn_tile_geom(n, c);
  BnTileStats a{};
  a.partial = partial; a.G = G; a.slice = n * c; a.acc_in = (G > 0 && accumulate) ? dout : nullptr; a.src = dout;
  // the masked gradient goes to the residual branch's buffer when there is one; else (slices: dout is this program's
  // own buffer) in place; a finished dout without a residual consumer is left alone and masked again in the dx pass
  a.dst = dres ? dres : (G > 0 ? dout : nullptr);
  a.x = y; a.y_out = y_out; a.mean = mean; a.invstd = invstd; a.n = n; a.c = c; a.tr = g.tr; a.tpart = (double*)ws;
  hipLaunchKernelGGL((bn_tile_stats_kernel<true>), dim3((unsigned)g.ntiles, (unsigned)g.ncb), dim3(256), 0, st, a);
  BnTileDx b{};
  b.tpart = (const double*)ws; b.ntiles = g.ntiles;
  b.o = BnBwdOut{dgamma, dbeta, nullptr, nullptr, n, (int)training, (int)dbn_accumulate};
  b.g = a.dst ? a.dst : dout;
  b.mask_src = a.dst ? nullptr : y_out;
  b.x = y; b.mean = mean; b.invstd = invstd; b.gamma = gamma; b.dx = dy; b.nvec = n * c / 4; b.c = c;
  hipLaunchKernelGGL(bn_tile_bwd_dx_kernel, dim3((unsigned)bn_apply_grid(b.nvec)), dim3(256), (size_t)2 * c * 4, st, b);
  USC_CHECK_LAUNCH("usc_bn_tile_backward");
  return USC_OK;
}

int usc_relu_fwd(const float* x, float* y, int64_t numel, usc_stream_t s) {
  USC_REQUIRE(numel >= 0, "usc_relu_fwd: bad size");
  if (numel == 0) return USC_OK;
  USC_REQUIRE(x && y, "usc_relu_fwd: null pointer");
  hipLaunchKernelGGL(relu_fwd_kernel, dim3(stream_grid(numel / 4 + 1, 256)), dim3(256), 0, as_stream(s), x, y, numel);
  USC_CHECK_LAUNCH("usc_relu_fwd");
  return USC_OK;
}
int usc_relu_bwd(const float* y, const float* dy, float* dx, int64_t numel, usc_stream_t s) {
  USC_REQUIRE(numel >= 0, "usc_relu_bwd: bad size");
  if (numel == 0) return USC_OK;
  USC_REQUIRE(y && dy && dx, "usc_relu_bwd: null pointer");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(stream_grid(numel / 4 + 1, 256)), dim3(256), 0, as_stream(s), y, dy, dx,
                     numel);
  USC_CHECK_LAUNCH("usc_relu_bwd");
  return USC_OK;
}

int usc_avgpool_down2(const float* in, int32_t c, const int32_t* nbr2, int64_t n_coarse, float* out, usc_stream_t s) {
  return usc_avgpool_down2_ex(in, c, c, nullptr, nbr2, n_coarse, out, nullptr, s);
}

int usc_avgpool_down2_ex(const float* in, int32_t c, int32_t ld, const int64_t* row_of, const int32_t* nbr2,
                         int64_t n_coarse, float* out, uint8_t* mask_out, usc_stream_t s) {
  USC_REQUIRE(c >= 1 && ld >= c && n_coarse >= 0, "usc_avgpool_down2: bad sizes");
  if (n_coarse == 0) return USC_OK;
  USC_REQUIRE(in && nbr2 && (out || mask_out), "usc_avgpool_down2: null pointer");
  if (c % 4 == 0 && ld % 4 == 0)
    hipLaunchKernelGGL((avgpool_down2_kernel<4>), dim3(stream_grid(n_coarse * (c / 4), 256)), dim3(256), 0,
                       as_stream(s), in, (int)c, (int)ld, row_of, nbr2, n_coarse, out, mask_out);
  else
    hipLaunchKernelGGL((avgpool_down2_kernel<1>), dim3(stream_grid(n_coarse * c, 256)), dim3(256), 0, as_stream(s), in,
                       (int)c, (int)ld, row_of, nbr2, n_coarse, out, mask_out);
  USC_CHECK_LAUNCH("usc_avgpool_down2");
  return USC_OK;
}

int usc_gather_rows(const float* src, int32_t c, const int64_t* idx, int64_t n, float* out, usc_stream_t s) {
  USC_REQUIRE(c >= 1 && n >= 0, "usc_gather_rows: bad sizes");
  if (n == 0) return USC_OK;
  USC_REQUIRE(src && idx && out, "usc_gather_rows: null pointer");
  if (c % 4 == 0)
    hipLaunchKernelGGL((gather_rows_kernel<4>), dim3(stream_grid(n * (c / 4), 256)), dim3(256), 0, as_stream(s), src,
                       (int)c, idx, n, out);
  else
    hipLaunchKernelGGL((gather_rows_kernel<1>), dim3(stream_grid(n * c, 256)), dim3(256), 0, as_stream(s), src, (int)c,
                       idx, n, out);
  USC_CHECK_LAUNCH("usc_gather_rows");
  return USC_OK;
}

int64_t usc_sample_keys_ws_bytes(int32_t n_scenes, int32_t K, int32_t q) {
  return (int64_t)n_scenes * ceil_div((int64_t)K, (int64_t)usc::kSampleRows) * ((q + 3) / 4 * 4);
}

int usc_sample_keys(const float* feats, int32_t c, const uint8_t* mask, int32_t q, const float* pos, int32_t p,
                    const int64_t* idx, int32_t n_scenes, int32_t K, const int32_t* n_valid, float* out_feats,
                    uint8_t* out_mask, float* out_pos, void* ws, int64_t ws_bytes, usc_stream_t s) {
  // feats == NULL with c == 0: mask rows only (the part of a pass's keys that depends on the queries); mask == NULL with
  // q == 0: feature / positional rows only (the part that does not: issued ahead of the decoder loop, on another stream)
  USC_REQUIRE(n_scenes >= 1 && n_scenes <= usc::kSampleMaxScenes && K >= 1 && c >= 0 && c % 4 == 0 && q >= 0 &&
                  c <= 512 && q <= usc::kSampleMaxQ && (pos == nullptr || (p >= 4 && p % 4 == 0 && p <= 512)) &&
                  (c > 0 || q > 0),
              "usc_sample_keys: unsupported sizes (scenes <= 16, queries <= 128, channel counts multiples of 4 up to 512)");
  USC_REQUIRE(idx && n_valid && (c == 0 || (feats && out_feats)) && (q == 0 || (mask && out_mask && ws)) &&
                  (pos == nullptr || out_pos),
              "usc_sample_keys: null pointer");
  if (c == 0) { feats = nullptr; out_feats = nullptr; }
  if (q == 0) { mask = nullptr; out_mask = nullptr; }
  usc::SampleArgs a{};
  a.feats = feats; a.mask = mask; a.pos = pos; a.idx = idx;
  a.out_feats = out_feats; a.out_mask = out_mask; a.out_pos = out_pos; a.part = (unsigned char*)ws;
  a.c = c; a.q = q; a.p = p; a.K = K;
  a.nblk = (int)ceil_div((int64_t)K, (int64_t)usc::kSampleRows);
  a.qs = (q + 3) / 4 * 4;
  USC_REQUIRE(q == 0 || ((uintptr_t)ws & 3) == 0, "usc_sample_keys: workspace must be 4-byte aligned");
  USC_REQUIRE(q == 0 || ws_bytes >= usc_sample_keys_ws_bytes(n_scenes, K, q), "usc_sample_keys: workspace too small");
  for (int b = 0; b < n_scenes; ++b) {
    USC_REQUIRE(n_valid[b] >= 1, "usc_sample_keys: a scene without rows");
    a.n_valid[b] = n_valid[b];
  }
  hipLaunchKernelGGL(usc::sample_keys_kernel, dim3((unsigned)a.nblk, (unsigned)n_scenes), dim3(256), 0, as_stream(s), a);
  if (q > 0)
    hipLaunchKernelGGL(usc::sample_fix_kernel, dim3((unsigned)a.nblk, (unsigned)n_scenes), dim3(256), 0, as_stream(s), a);
  USC_CHECK_LAUNCH("usc_sample_keys");
  return USC_OK;
}

int usc_scatter_rows_unique(const float* src, int32_t c, const int64_t* idx, int64_t n, float* dst, usc_stream_t s) {
  USC_REQUIRE(c >= 1 && n >= 0, "usc_scatter_rows_unique: bad sizes");
  if (n == 0) return USC_OK;
  USC_REQUIRE(src && idx && dst, "usc_scatter_rows_unique: null pointer");
  if (c % 4 == 0)
    hipLaunchKernelGGL((usc::scatter_rows_kernel<4>), dim3(stream_grid(n * (c / 4), 256)), dim3(256), 0, as_stream(s), src,
                       (int)c, idx, n, dst);
  else
    hipLaunchKernelGGL((usc::scatter_rows_kernel<1>), dim3(stream_grid(n * c, 256)), dim3(256), 0, as_stream(s), src, (int)c,
                       idx, n, dst);
  USC_CHECK_LAUNCH("usc_scatter_rows_unique");
  return USC_OK;
}

int usc_scatter_rows_unique_add(const float* src, int32_t c, const int64_t* idx, int64_t n, float* dst, usc_stream_t s) {
  USC_REQUIRE(c >= 1 && n >= 0, "usc_scatter_rows_unique_add: bad sizes");
  if (n == 0) return USC_OK;
  USC_REQUIRE(src && idx && dst, "usc_scatter_rows_unique_add: null pointer");
  if (c % 4 == 0)
    hipLaunchKernelGGL((usc::scatter_rows_rmw_kernel<4>), dim3(stream_grid(n * (c / 4), 256)), dim3(256), 0, as_stream(s),
                       src, (int)c, idx, n, dst);
  else
    hipLaunchKernelGGL((usc::scatter_rows_rmw_kernel<1>), dim3(stream_grid(n * c, 256)), dim3(256), 0, as_stream(s), src,
                       (int)c, idx, n, dst);
  USC_CHECK_LAUNCH("usc_scatter_rows_unique_add");
  return USC_OK;
}

int usc_scatter_add_rows(const float* src, int32_t c, const int64_t* idx, int64_t n, float* dst, usc_stream_t s) {
  USC_REQUIRE(c >= 1 && n >= 0, "usc_scatter_add_rows: bad sizes");
  if (n == 0) return USC_OK;
  USC_REQUIRE(src && idx && dst, "usc_scatter_add_rows: null pointer");
  if (c % 4 == 0)
    hipLaunchKernelGGL((scatter_add_rows_kernel<4>), dim3(stream_grid(n * (c / 4), 256)), dim3(256), 0, as_stream(s),
                       src, (int)c, idx, n, dst);
  else
    hipLaunchKernelGGL((scatter_add_rows_kernel<1>), dim3(stream_grid(n * c, 256)), dim3(256), 0, as_stream(s), src,
                       (int)c, idx, n, dst);
  USC_CHECK_LAUNCH("usc_scatter_add_rows");
  return USC_OK;
}

static int64_t seg_tiles(int64_t n) { return n > 0 ? ceil_div(n, kSegTile) : 1; }
int64_t usc_segment_csr_ws_bytes(int64_t n, int64_t S) {
  // hist i32[T][S] | counts i32[S] | scan ws
  return align_up(seg_tiles(n) * S * 4, 16) + align_up(S * 4, 16) + scan_ws_bytes(S + 1);
}

int usc_segment_csr(const int64_t* seg, int64_t n, int64_t S, int64_t* order, int64_t* seg_off, void* ws,
                    int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(n >= 0 && S >= 1, "usc_segment_csr: bad sizes");
  USC_REQUIRE(seg_off && ws && (n == 0 || (seg && order)), "usc_segment_csr: null pointer");
  USC_REQUIRE(ws_bytes >= usc_segment_csr_ws_bytes(n, S), "usc_segment_csr: workspace too small");
  hipStream_t st = as_stream(s);
  const int64_t T = seg_tiles(n);
  char* w = (char*)ws;
  int32_t* hist = (int32_t*)w;
  int32_t* counts = (int32_t*)(w + align_up(T * S * 4, 16));
  void* scan_ws = w + align_up(T * S * 4, 16) + align_up(S * 4, 16);
  (void)hipMemsetAsync(hist, 0, (size_t)(T * S * 4), st);
  if (n > 0) hipLaunchKernelGGL(seg_hist_kernel, dim3((unsigned)T), dim3(256), 0, st, seg, n, S, hist);
  hipLaunchKernelGGL(seg_tilescan_kernel, dim3((unsigned)ceil_div(S, 256)), dim3(256), 0, st, hist, T, S, counts);
  // seg_off[0..S] = exclusive scan of counts (entry S = total): scan S+1 values with counts[S] := 0
  struct CountValPad {
    const int32_t* counts; int64_t S;
    __device__ int operator()(int64_t i) const { return i < S ? counts[i] : 0; }
  };
  CountValPad f{counts, S};
  SegOffEmit e{seg_off};
  device_exclusive_scan(f, e, S + 1, scan_ws, st);
  if (n > 0) hipLaunchKernelGGL(seg_place_kernel, dim3((unsigned)T), dim3(64), 0, st, seg, n, S, hist, seg_off, order);
  USC_CHECK_LAUNCH("usc_segment_csr");
  return USC_OK;
}

int usc_segment_mean_fwd(const float* src, int32_t c, const int64_t* order, const int64_t* seg_off, int64_t S,
                         float* out, usc_stream_t s) {
  USC_REQUIRE(c >= 1 && S >= 0, "usc_segment_mean_fwd: bad sizes");
  if (S == 0) return USC_OK;
  USC_REQUIRE(src && order && seg_off && out, "usc_segment_mean_fwd: null pointer");
  if (c % 4 == 0 && c <= 128)
    hipLaunchKernelGGL(segment_mean_fwd_vec_kernel, dim3((unsigned)S), dim3(256), 0, as_stream(s), src, (int)c, order,
                       seg_off, S, out);
  else
    hipLaunchKernelGGL(segment_mean_fwd_kernel, dim3((unsigned)ceil_div(S, 4)), dim3(256), 0, as_stream(s), src,
                       (int)c, order, seg_off, S, 0, out, (int64_t*)nullptr);
  USC_CHECK_LAUNCH("usc_segment_mean_fwd");
  return USC_OK;
}

int usc_segment_mean_nonzero(const float* feats, int32_t d, const int64_t* order, const int64_t* seg_off, int64_t S,
                             float* out, int64_t* nonzero_cnt, usc_stream_t s) {
  USC_REQUIRE(d >= 1 && S >= 0, "usc_segment_mean_nonzero: bad sizes");
  if (S == 0) return USC_OK;
  USC_REQUIRE(feats && order && seg_off && out, "usc_segment_mean_nonzero: null pointer");
  hipLaunchKernelGGL(segment_mean_fwd_kernel, dim3((unsigned)ceil_div(S, 4)), dim3(256), 0, as_stream(s), feats, (int)d,
                     order, seg_off, S, 1, out, nonzero_cnt);
  USC_CHECK_LAUNCH("usc_segment_mean_nonzero");
  return USC_OK;
}

int usc_segment_max_nonzero(const float* feats, int32_t d, const int64_t* order, const int64_t* seg_off, int64_t S,
                            float* out, int64_t* nonzero_cnt, usc_stream_t s) {
  USC_REQUIRE(d >= 1 && S >= 0, "usc_segment_max_nonzero: bad sizes");
  if (S == 0) return USC_OK;
  USC_REQUIRE(feats && order && seg_off && out, "usc_segment_max_nonzero: null pointer");
  hipLaunchKernelGGL(segment_mean_fwd_kernel, dim3((unsigned)ceil_div(S, 4)), dim3(256), 0, as_stream(s), feats, (int)d,
                     order, seg_off, S, 3, out, nonzero_cnt);
  USC_CHECK_LAUNCH("usc_segment_max_nonzero");
  return USC_OK;
}

int usc_segment_mean_bwd(const float* dout, int32_t c, const int64_t* seg, const int64_t* seg_off, int64_t n,
                         float* dsrc, usc_stream_t s) {
  USC_REQUIRE(c >= 1 && n >= 0, "usc_segment_mean_bwd: bad sizes");
  if (n == 0) return USC_OK;
  USC_REQUIRE(dout && seg && seg_off && dsrc, "usc_segment_mean_bwd: null pointer");
  if (c % 4 == 0)
    hipLaunchKernelGGL((segment_mean_bwd_kernel<4>), dim3(stream_grid(n * (c / 4), 256)), dim3(256), 0, as_stream(s),
                       dout, (int)c, seg, seg_off, n, dsrc);
  else
    hipLaunchKernelGGL((segment_mean_bwd_kernel<1>), dim3(stream_grid(n * c, 256)), dim3(256), 0, as_stream(s), dout,
                       (int)c, seg, seg_off, n, dsrc);
  USC_CHECK_LAUNCH("usc_segment_mean_bwd");
  return USC_OK;
}

int usc_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int64_t step, usc_stream_t s) {
  USC_REQUIRE(n >= 0 && step >= 1, "usc_adamw_step: bad sizes");
  if (n == 0) return USC_OK;
  USC_REQUIRE(param && grad && exp_avg && exp_avg_sq, "usc_adamw_step: null pointer");
  USC_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
              "usc_adamw_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(adamw_kernel, dim3(stream_grid(n4 > 0 ? n4 : 1, 256)), dim3(256), 0, as_stream(s), param, grad,
                     exp_avg, exp_avg_sq, n4, n, lr, beta1, beta2, eps, weight_decay, step_size, bc2_sqrt);
  USC_CHECK_LAUNCH("usc_adamw_step");
  return USC_OK;
}

}  // extern "C"
