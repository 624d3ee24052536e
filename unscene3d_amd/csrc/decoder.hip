// decoder.hip — small-row kernels of the 100-query mask decoder (gfx950).
// LayerNorm forward / backward over [rows, d] with rows ~ 100 and d = 128: the stock backward spends 31 us per
// call in its gamma/beta reduction kernel (57 calls per training step); here the whole backward is one launch.
// Reference: nn.LayerNorm in models/mask3d.py:174,515,572,627 (decoder_norm and the post-norms of the
// self-attention / cross-attention / FFN layers).
#include "common.h"

namespace usc {
namespace {

constexpr int kLnMaxPerLane = 8;    // d <= 512

// one wave per row; lane l holds columns l, l + 64, ...
// res (optional): the normalised row is x + res (post-norm residual, models/mask3d.py:523,543), written to sum_out
// for the backward pass
template <int PER>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int64_t rows, int d, float eps,
                                                           float* __restrict__ y, float* __restrict__ sum_out,
                                                           float* __restrict__ mean, float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float v[PER];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    v[j] = x[r * d + lane + 64 * j];
    if (res) {
      v[j] += res[r * d + lane + 64 * j];
      sum_out[r * d + lane + 64 * j] = v[j];
    }
    s += v[j];
  }
  const float mu = wave_reduce_addf(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < PER; ++j) { const float t = v[j] - mu; q += t * t; }
  const float rs = rsqrtf(wave_reduce_addf(q) / (float)d + eps);
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int c = lane + 64 * j;
    y[r * d + c] = (v[j] - mu) * rs * gamma[c] + beta[c];
  }
  if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
}

// block b owns rows [b*R, (b+1)*R): dx for its rows and the partial column sums of dy*xhat / dy in wave order
template <int PER>
__global__ __launch_bounds__(1024) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, int64_t rows, int d,
                                                            int64_t rows_per_block, int accumulate,
                                                            const float* __restrict__ dx_add, float* __restrict__ dx,
                                                            float* __restrict__ part /* [G][2][d] or dgamma/dbeta */,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[16][2][64 * PER];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float g[PER], dg[PER], db[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) { g[j] = gamma[lane + 64 * j]; dg[j] = 0.f; db[j] = 0.f; }
  // Few rows (the decoder's 100 queries: ONE workgroup, 7 rows per wave): all of a wave's rows are fetched before the
  // first is used — the row loop below is one dependent load round trip per row (7 x ~1.2 us of a 9 us launch, 49
  // launches per training step).  Same arithmetic and the same per-wave row order: identical results.
  constexpr int RW = (PER <= 2) ? 8 : 4;
  if (r1 - r0 <= 16 * RW) {
    float xv[RW][PER], dv[RW][PER], av[RW][PER], mu[RW], rs[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) {
      const int64_t r = r0 + wave + 16 * q;
      const int64_t rr = r < r1 ? r : r0;
      mu[q] = mean[rr];
      rs[q] = rstd[rr];
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        xv[q][j] = x[rr * d + lane + 64 * j];
        dv[q][j] = dy[rr * d + lane + 64 * j];
        av[q][j] = dx_add ? dx_add[rr * d + lane + 64 * j] : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < RW; ++q) {
      const int64_t r = r0 + wave + 16 * q;
      if (r < r1) {
        float xh[PER], gy[PER];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          xh[j] = (xv[q][j] - mu[q]) * rs[q];
          gy[j] = dv[q][j] * g[j];
          dg[j] += dv[q][j] * xh[j];
          db[j] += dv[q][j];
          s1 += gy[j];
          s2 += gy[j] * xh[j];
        }
        s1 = wave_reduce_addf(s1) / (float)d;
        s2 = wave_reduce_addf(s2) / (float)d;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          const float v = rs[q] * (gy[j] - s1 - xh[j] * s2);
          dx[r * d + lane + 64 * j] = dx_add ? v + av[q][j] : v;
        }
      }
    }
  } else
  for (int64_t r = r0 + wave; r < r1; r += 16) {
    const float mu = mean[r], rs = rstd[r];
    float xh[PER], gy[PER];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int c = lane + 64 * j;
      const float dyv = dy[r * d + c];
      xh[j] = (x[r * d + c] - mu) * rs;
      gy[j] = dyv * g[j];
      dg[j] += dyv * xh[j];
      db[j] += dyv;
      s1 += gy[j];
      s2 += gy[j] * xh[j];
    }
    s1 = wave_reduce_addf(s1) / (float)d;
    s2 = wave_reduce_addf(s2) / (float)d;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const float v = rs * (gy[j] - s1 - xh[j] * s2);
      // dx_add: the gradient that reaches x around the norm (a second consumer of x), summed here instead of by autograd
      dx[r * d + lane + 64 * j] = dx_add ? v + dx_add[r * d + lane + 64 * j] : v;
    }
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) { red[wave][0][lane + 64 * j] = dg[j]; red[wave][1][lane + 64 * j] = db[j]; }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * d; c += 1024) {
    const int which = c / d, col = c - which * d;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w][which][col];
    if (gridDim.x == 1) {
      float* dst = (which == 0 ? dgamma : dbeta) + col;
      *dst = accumulate ? *dst + s : s;
    } else {
      part[((int64_t)blockIdx.x * 2 + which) * d + col] = s;
    }
  }
}

__global__ __launch_bounds__(256) void layernorm_bwd_final_kernel(const float* __restrict__ part, int G, int d,
                                                                 int accumulate, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= 2 * d) return;
  const int which = c / d, col = c - which * d;
  float s = 0.f;
  for (int b = 0; b < G; ++b) s += part[((int64_t)b * 2 + which) * d + col];
  float* dst = (which == 0 ? dgamma : dbeta) + col;
  *dst = accumulate ? *dst + s : s;
}

constexpr int64_t kLnRowsPerBlock = 1024;

}  // namespace
}  // namespace usc

using namespace usc;

extern "C" {

int usc_layernorm_fwd(const float* x, const float* gamma, const float* beta, int64_t rows, int32_t d, float eps, float* y,
                      float* mean, float* rstd, usc_stream_t s) {
  return usc_add_layernorm_fwd(x, nullptr, gamma, beta, rows, d, eps, y, nullptr, mean, rstd, s);
}

int usc_add_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, int64_t rows, int32_t d,
                          float eps, float* y, float* sum_out, float* mean, float* rstd, usc_stream_t s) {
  USC_REQUIRE(rows >= 0 && d >= 64 && d % 64 == 0 && d <= 64 * kLnMaxPerLane, "usc_layernorm_fwd: d must be a multiple of 64, <= 512");
  if (rows == 0) return USC_OK;
  USC_REQUIRE(x && gamma && beta && y && mean && rstd && (!res || sum_out), "usc_layernorm_fwd: null pointer");
  const dim3 grid((unsigned)ceil_div(rows, 4));
  hipStream_t st = as_stream(s);
#define USC_LN_F(P) hipLaunchKernelGGL((layernorm_fwd_kernel<P>), grid, dim3(256), 0, st, x, res, gamma, beta, rows, (int)d, eps, y, sum_out, mean, rstd)
  switch (d / 64) {
    case 1: USC_LN_F(1); break;  case 2: USC_LN_F(2); break;  case 3: USC_LN_F(3); break;  case 4: USC_LN_F(4); break;
    case 6: USC_LN_F(6); break;  case 8: USC_LN_F(8); break;
    default: USC_REQUIRE(false, "usc_layernorm_fwd: unsupported d (64*{1,2,3,4,6,8})");
  }
#undef USC_LN_F
  USC_CHECK_LAUNCH("usc_layernorm_fwd");
  return USC_OK;
}

int64_t usc_layernorm_bwd_ws_bytes(int64_t rows, int32_t d) {
  const int64_t G = ceil_div(rows, kLnRowsPerBlock);
  return G > 1 ? G * 2 * d * 4 : 0;
}

int usc_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                      int64_t rows, int32_t d, float* dx, float* dgamma, float* dbeta, int32_t accumulate, void* ws,
                      int64_t ws_bytes, usc_stream_t s) {
  return usc_layernorm_bwd_ex(dy, x, mean, rstd, gamma, rows, d, nullptr, dx, dgamma, dbeta, accumulate, ws, ws_bytes, s);
}

int usc_layernorm_bwd_ex(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                         int64_t rows, int32_t d, const float* dx_add, float* dx, float* dgamma, float* dbeta,
                         int32_t accumulate, void* ws, int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(rows >= 1 && d >= 64 && d % 64 == 0 && d <= 64 * kLnMaxPerLane, "usc_layernorm_bwd: bad sizes");
  USC_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta, "usc_layernorm_bwd: null pointer");
  const int64_t G = ceil_div(rows, kLnRowsPerBlock);
  USC_REQUIRE(G == 1 || (ws && ws_bytes >= G * 2 * d * 4), "usc_layernorm_bwd: workspace too small");
  hipStream_t st = as_stream(s);
#define USC_LN_B(P) hipLaunchKernelGGL((layernorm_bwd_kernel<P>), dim3((unsigned)G), dim3(1024), 0, st, dy, x, mean, rstd, gamma, rows, (int)d, kLnRowsPerBlock, (int)accumulate, dx_add, dx, (float*)ws, dgamma, dbeta)
  switch (d / 64) {
    case 1: USC_LN_B(1); break;  case 2: USC_LN_B(2); break;  case 3: USC_LN_B(3); break;  case 4: USC_LN_B(4); break;
    case 6: USC_LN_B(6); break;  case 8: USC_LN_B(8); break;
    default: USC_REQUIRE(false, "usc_layernorm_bwd: unsupported d (64*{1,2,3,4,6,8})");
  }
#undef USC_LN_B
  if (G > 1)
    hipLaunchKernelGGL(layernorm_bwd_final_kernel, dim3((unsigned)ceil_div(2 * d, 256)), dim3(256), 0, st,
                       (const float*)ws, (int)G, (int)d, (int)accumulate, dgamma, dbeta);
  USC_CHECK_LAUNCH("usc_layernorm_bwd");
  return USC_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------
// Linear layers with a handful of rows (the 100 queries): y = x W^T + b and its two gradients on the f32
// matrix cores, one workgroup per 32x32 output tile, everything in one launch each.  The library GEMM picks a
// 128x128 macro tile for these shapes (one workgroup walking K): ~12 us per call, ~25 calls per decoder pass.
// Reference: nn.Linear / nn.MultiheadAttention projections of models/mask3d.py:491-651 (attention layers, FFN)
// and :70-72 (mask_embed_head).
namespace usc {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__device__ inline int acc_row16(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// All three products: a 256-thread workgroup per 32x32 output tile; the four waves split the REDUCTION dimension
// (one wave per tile was latency bound: 32 dependent load->MFMA rounds for the 1024-wide FFN), partial tiles are
// summed through LDS in wave order (deterministic), each wave finalising four of the sixteen accumulator rows.
__device__ inline void tile_reduce_store(f32x16 acc, float (*red)[16][64], int wave, int lane, float* __restrict__ out,
                                         int64_t ld, int row0, int col0, int max_row, const float* __restrict__ bias,
                                         int accumulate = 0, int relu = 0, const float* __restrict__ add = nullptr,
                                         const float* __restrict__ add2 = nullptr, float* __restrict__ out_b = nullptr,
                                         int zero_rows_to = 0) {
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  const int i = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 4 * wave + q;
    const float v = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane];
    const int row = row0 + acc_row16(r, h);
    if (row < max_row) {
      float* dst = out + (int64_t)row * ld + col0 + i;
      float o = v + (bias ? bias[col0 + i] : 0.f) + (accumulate ? *dst : 0.f);
      if (add) o += add[(int64_t)row * ld + col0 + i];
      if (out_b) out_b[(int64_t)row * ld + col0 + i] = o;      // the sum without add2 (second consumer of the product)
      if (add2) o += add2[(int64_t)row * ld + col0 + i];
      *dst = relu ? fmaxf(o, 0.f) : o;
    } else if (row < zero_rows_to) {
      out[(int64_t)row * ld + col0 + i] = 0.f;       // padding rows of a table the caller wants zero-extended
    }
  }
}

// y[M,N] = x[M,K] W[N,K]^T (+ b)      grid (N/32, ceil(M/32)); K % 32 == 0
// x2 (optional): the input rows are x + x2 (positional encodings, models/mask3d.py:485,517); relu: y = max(y, 0)
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ x2,
                                                        const float* __restrict__ W, const float* __restrict__ b, int M,
                                                        int N, int K, int relu, float* __restrict__ y, int Mpad,
                                                        int x2_cols, int split) {
  // x2_cols: the positional term enters output columns < x2_cols only; split > 0: the output is [N/split][M][split]
  // (column block j = n / split is its own contiguous matrix) — the q, k, v projections of a self-attention block in
  // one launch: q = (x + pos) Wq, k = (x + pos) Wk, v = x Wv
  __shared__ float red[4][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int m = m0 + i;
  const float* xa = x + (int64_t)(m < M ? m : 0) * K + 4 * h;
  const float* xa2 = (x2 && n0 < x2_cols) ? x2 + (int64_t)(m < M ? m : 0) * K + 4 * h : nullptr;
  const float* wb = W + (int64_t)(n0 + i) * K + 4 * h;
  const float keep = m < M ? 1.f : 0.f;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nchunk = K >> 5;
  for (int c = wave; c < nchunk; c += 4) {
    const int k0 = c * 32;
    float4 a[4], bb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { a[t] = *reinterpret_cast<const float4*>(xa + k0 + 8 * t); bb[t] = *reinterpret_cast<const float4*>(wb + k0 + 8 * t); }
    if (xa2) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 e = *reinterpret_cast<const float4*>(xa2 + k0 + 8 * t);
        a[t].x += e.x; a[t].y += e.y; a[t].z += e.z; a[t].w += e.w;
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float av[4] = {a[t].x * keep, a[t].y * keep, a[t].z * keep, a[t].w * keep};
      const float bv[4] = {bb[t].x, bb[t].y, bb[t].z, bb[t].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = MFMA32(av[j], bv[j], acc);
    }
  }
  if (split > 0) {
    const int j = n0 / split;
    tile_reduce_store(acc, red, wave, lane, y + (int64_t)j * M * split, split, m0, n0 - j * split, M,
                      b ? b + j * split : nullptr, 0, relu);
    return;
  }
  tile_reduce_store(acc, red, wave, lane, y, N, m0, n0, M, b, 0, relu, nullptr, nullptr, nullptr, Mpad);
}

// dx[M,K] = dy[M,N] W[N,K]   for the 32x32 tile (c0, m0); N % 32 == 0.  The four waves split N; every wave keeps the
// NEXT chunk's operands in flight while the matrix cores work on the current one (one wave per SIMD: nothing else
// hides the ~1 us load latency, and with 8 chunks per wave for the 1024-wide FFN the loop was 8 dependent round trips).
// ymask (optional): the layer's forward output after its fused ReLU — dy counts only where it is > 0;
// dx_add (optional, [M,K]): added to the result (the other gradient path into the same input);
// dx_b (optional, [M,K]): also receives dy W + dx_add, while dx receives dy W + dx_add + dx_add2 — the product feeds
// two gradients (the layer input's, which also has a residual path, and the positional term's, which has not).
template <bool EX>
__device__ inline void linear_dx_tile(const float* __restrict__ dy, const float* __restrict__ ymask,
                                      const float* __restrict__ W, int M, int N, int K, float* __restrict__ dx,
                                      const float* __restrict__ dx_add, const float* __restrict__ dx_add2,
                                      float* __restrict__ dx_b, int c0, int m0, float (*red)[16][64]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  const int m = m0 + i;
  const float* ya = dy + (int64_t)(m < M ? m : 0) * N + 4 * h;
  const float* ym = (EX && ymask) ? ymask + (int64_t)(m < M ? m : 0) * N + 4 * h : nullptr;
  const float* wb = W + (int64_t)(4 * h) * K + c0 + i;
  const float keep = m < M ? 1.f : 0.f;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nchunk = N >> 5;
  float4 a[4], an[4];
  float bv[16], bn[16];
  auto load = [&](int c, float4* aa, float* bb) __attribute__((always_inline)) {
    const int n0 = c * 32;
#pragma unroll
    for (int t = 0; t < 4; ++t) aa[t] = *reinterpret_cast<const float4*>(ya + n0 + 8 * t);
    if (EX && ym) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 o = *reinterpret_cast<const float4*>(ym + n0 + 8 * t);
        aa[t].x = o.x > 0.f ? aa[t].x : 0.f; aa[t].y = o.y > 0.f ? aa[t].y : 0.f;
        aa[t].z = o.z > 0.f ? aa[t].z : 0.f; aa[t].w = o.w > 0.f ? aa[t].w : 0.f;
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) bb[4 * t + j] = wb[(int64_t)(n0 + 8 * t + j) * K];
  };
  int c = wave;
  if (c < nchunk) load(c, an, bn);
  for (; c < nchunk; c += 4) {
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = an[t];
#pragma unroll
    for (int q = 0; q < 16; ++q) bv[q] = bn[q];
    if (c + 4 < nchunk) load(c + 4, an, bn);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float av[4] = {a[t].x * keep, a[t].y * keep, a[t].z * keep, a[t].w * keep};
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = MFMA32(av[j], bv[4 * t + j], acc);
    }
  }
  tile_reduce_store(acc, red, wave, lane, dx, K, m0, c0, M, nullptr, 0, 0, EX ? dx_add : nullptr,
                    EX ? dx_add2 : nullptr, EX ? dx_b : nullptr);
}

// dW[N,K] = dy[M,N]^T x[M,K],  db[N] = sum_m dy[m][N]   for the tile (c0, n0); the tiles with c0 == 0 also write db
// ymask as above; x2 (optional): the layer's input rows were x + x2
template <bool EX>
__device__ inline void linear_dw_tile(const float* __restrict__ dy, const float* __restrict__ ymask,
                                      const float* __restrict__ x, const float* __restrict__ x2, int M, int N, int K,
                                      int accumulate, float* __restrict__ dW, float* __restrict__ db, int c0, int n0,
                                      float (*red)[16][64], float (*bred)[32]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  for (int mb = wave * 32; mb < M; mb += 128) {
    float av[16], bv[16];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = mb + 8 * t + 4 * h + j;
        const bool ok = m < M;
        av[4 * t + j] = ok ? dy[(int64_t)m * N + n0 + i] : 0.f;
        bv[4 * t + j] = ok ? x[(int64_t)m * K + c0 + i] : 0.f;
      }
    if (EX && ymask) {        // (uniform branches, all 16 loads of a block in flight together)
      float ov[16];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = mb + 8 * t + 4 * h + j;
          ov[4 * t + j] = m < M ? ymask[(int64_t)m * N + n0 + i] : 1.f;
        }
#pragma unroll
      for (int q = 0; q < 16; ++q) av[q] = ov[q] > 0.f ? av[q] : 0.f;
    }
    if (EX && x2) {
      float ev[16];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = mb + 8 * t + 4 * h + j;
          ev[4 * t + j] = m < M ? x2[(int64_t)m * K + c0 + i] : 0.f;
        }
#pragma unroll
      for (int q = 0; q < 16; ++q) bv[q] += ev[q];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc = MFMA32(av[q], bv[q], acc); bsum += av[q]; }
  }
  const bool want_b = db && c0 == 0;
  if (want_b) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (h == 0) bred[wave][i] = bsum;
  }
  tile_reduce_store(acc, red, wave, lane, dW, K, n0, c0, N, nullptr, accumulate);   // contains the __syncthreads
  if (want_b && threadIdx.x < 32) {
    const float v = bred[0][threadIdx.x] + bred[1][threadIdx.x] + bred[2][threadIdx.x] + bred[3][threadIdx.x];
    db[n0 + threadIdx.x] = accumulate ? db[n0 + threadIdx.x] + v : v;
  }
}

// Both gradients of a few-row linear layer in ONE launch: workgroups [0, dx_tiles) take the input-gradient tiles,
// the rest the weight-gradient tiles (they are independent; as two launches the pair cost 9.4 + 4.5 us of a
// latency-bound decoder pass, ~150 pairs per training step).
template <bool EX>
__global__ __launch_bounds__(256) void linear_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ymask,
                                                        const float* __restrict__ x, const float* __restrict__ x2,
                                                        const float* __restrict__ W, int M, int N, int K, int dx_tiles,
                                                        int accumulate, float* __restrict__ dx,
                                                        const float* __restrict__ dx_add,
                                                        const float* __restrict__ dx_add2, float* __restrict__ dx_b,
                                                        float* __restrict__ dW, float* __restrict__ db) {
  // EX = false: the plain layer (no ymask / x2 / dx_add) — the extra operands cost 50 % in this latency-bound kernel
  // when merely tested per element, so the plain form is its own instantiation
  __shared__ float red[4][16][64];
  __shared__ float bred[4][32];
  const int kt = K / 32;
  int b = blockIdx.x;
  if (b < dx_tiles) {
    linear_dx_tile<EX>(dy, ymask, W, M, N, K, dx, dx_add, dx_add2, dx_b, (b % kt) * 32, (b / kt) * 32, red);
  } else {
    b -= dx_tiles;
    linear_dw_tile<EX>(dy, ymask, x, x2, M, N, K, accumulate, dW, db, (b % kt) * 32, (b / kt) * 32, red, bred);
  }
}


// ---- the three input projections of a self-attention block, backwards, in ONE launch (three launches before) -------
//   dy3 [3][M][E] = dq | dk | dv,   W [3E][E] (packed in_proj_weight),   x [M][E],   pos [M][E] or NULL
//   dx_b = dq Wq + dk Wk                  gradient of the positional term (q = k = (x + pos) W)        (optional)
//   dx   = dq Wq + dk Wk + dv Wv + dres   gradient of x; dres (optional): the block's residual path
//   dW[jE + n][c] (+)= sum_m dy_j[m][n] (x + [j < 2] pos)[m][c],   db[jE + n] (+)= sum_m dy_j[m][n]
// Workgroups [0, dx_tiles) take the input-gradient tiles (two accumulators: the q/k part and the v part, so that both
// sums come out of one pass over the 3E-long reduction), the rest the weight-gradient tiles of the three matrices.
__device__ inline void tile_reduce4(f32x16 acc, float (*red)[16][64], int wave, int lane, float (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 4 * wave + q;
    out[q] = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane];
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void qkv_bwd_kernel(const float* __restrict__ dy3, const float* __restrict__ x,
                                                     const float* __restrict__ pos, const float* __restrict__ W, int M,
                                                     int E, int dx_tiles, int accumulate, float* __restrict__ dx,
                                                     float* __restrict__ dx_b, const float* __restrict__ dres,
                                                     float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float red[4][16][64];
  __shared__ float bred[4][32];
  const int kt = E / 32;
  int b = blockIdx.x;
  if (b >= dx_tiles) {
    b -= dx_tiles;
    const int c0 = (b % kt) * 32, n0 = (b / kt) * 32, j = n0 / E;
    linear_dw_tile<true>(dy3 + (int64_t)j * M * E, nullptr, x, j < 2 ? pos : nullptr, M, E, E, accumulate,
                         dW + (int64_t)j * E * E, db ? db + j * E : nullptr, c0, n0 - j * E, red, bred);
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  const int c0 = (b % kt) * 32, m0 = (b / kt) * 32;
  const int m = m0 + i;
  const float keep = m < M ? 1.f : 0.f;
  f32x16 acc_qk, acc_v;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_qk[r] = acc_v[r] = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float* ya = dy3 + ((int64_t)j * M + (m < M ? m : 0)) * E + 4 * h;
    const float* wb = W + ((int64_t)j * E + 4 * h) * E + c0 + i;
    for (int c = wave; c < kt; c += 4) {
      const int n0 = c * 32;
      float4 a[4];
      float bv[16];
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const float4*>(ya + n0 + 8 * t);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) bv[4 * t + jj] = wb[(int64_t)(n0 + 8 * t + jj) * E];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float av[4] = {a[t].x * keep, a[t].y * keep, a[t].z * keep, a[t].w * keep};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          if (j < 2) acc_qk = MFMA32(av[jj], bv[4 * t + jj], acc_qk);
          else acc_v = MFMA32(av[jj], bv[4 * t + jj], acc_v);
        }
      }
    }
  }
  float qk[4], vv[4];
  tile_reduce4(acc_qk, red, wave, lane, qk);
  tile_reduce4(acc_v, red, wave, lane, vv);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = m0 + acc_row16(4 * wave + q, h);
    if (row < M) {
      const int64_t o = (int64_t)row * E + c0 + i;
      if (dx_b) dx_b[o] = qk[q];
      float t = qk[q] + vv[q];
      if (dres) t += dres[o];
      dx[o] = t;
    }
  }
}

// Column sums of a many-row table (the bias gradient of a linear layer over 3 200 / 12 800 sampled voxels), in a fixed
// order: 64-row partial sums, then per column 16 lanes add the partials ascending and their sums are added ascending.  No atomics, no
// last-block semaphores: the same bits on every launch and inside captured graphs.
constexpr int kColSumRows = 64;
__global__ __launch_bounds__(256) void col_sum_partial_kernel(const float* __restrict__ x, int64_t n, int c,
                                                             float* __restrict__ partial) {
  // 64 columns x 4 row lanes; a lane adds its 16 rows in four independent chains (rows r, r+16, r+32, r+48 of the
  // block), combined in a fixed order
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + tx;
  const int64_t r0 = (int64_t)blockIdx.x * kColSumRows;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (col < c) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = r0 + 16 * u + 4 * q + ty;
        if (r < n) a[u] += x[r * c + col];
      }
  }
  red[ty][tx] = (a[0] + a[1]) + (a[2] + a[3]);
  __syncthreads();
  if (ty == 0 && col < c) partial[(int64_t)blockIdx.x * c + col] = ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx];
}

__global__ __launch_bounds__(256) void col_sum_final_kernel(const float* __restrict__ partial, int64_t parts, int c,
                                                           int accumulate, float* __restrict__ out) {
  // 16 columns x 16 lanes; lane l adds partials l, l+16, ... ascending, then the 16 lane sums ascending
  __shared__ float red[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + tx;
  float s = 0.f;
  if (col < c)
    for (int64_t p = ty; p < parts; p += 16) s += partial[p * c + col];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && col < c) {
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) t += red[l][tx];
    out[col] = accumulate ? out[col] + t : t;
  }
}

}  // namespace
}  // namespace usc

extern "C" {

int usc_linear_fwd(const float* x, const float* W, const float* b, int32_t M, int32_t N, int32_t K, float* y,
                   usc_stream_t s) {
  return usc_linear_fwd_ex(x, nullptr, W, b, M, N, K, 0, y, s);
}

int usc_linear_fwd_ex(const float* x, const float* x2, const float* W, const float* b, int32_t M, int32_t N, int32_t K,
                      int32_t relu, float* y, usc_stream_t s) {
  return usc_linear_fwd_pad(x, x2, W, b, M, N, K, relu, y, M, s);
}

int usc_linear_fwd_pad(const float* x, const float* x2, const float* W, const float* b, int32_t M, int32_t N, int32_t K,
                       int32_t relu, float* y, int32_t M_pad, usc_stream_t s) {
  USC_REQUIRE(M >= 1 && N >= 32 && N % 32 == 0 && K >= 32 && K % 32 == 0, "usc_linear_fwd: N, K must be multiples of 32");
  USC_REQUIRE(M_pad >= M, "usc_linear_fwd_pad: M_pad < M");
  USC_REQUIRE(x && W && y, "usc_linear_fwd: null pointer");
  hipLaunchKernelGGL(usc::linear_fwd_kernel, dim3(N / 32, (M_pad + 31) / 32), dim3(256), 0, usc::as_stream(s), x, x2, W, b,
                     (int)M, (int)N, (int)K, (int)relu, y, (int)M_pad, (int)N, 0);
  USC_CHECK_LAUNCH("usc_linear_fwd");
  return USC_OK;
}

int usc_linear_fwd_split(const float* x, const float* x2, const float* W, const float* b, int32_t M, int32_t N, int32_t K,
                         int32_t x2_cols, int32_t split_cols, float* y, usc_stream_t s) {
  USC_REQUIRE(M >= 1 && N >= 32 && N % 32 == 0 && K >= 32 && K % 32 == 0, "usc_linear_fwd: N, K must be multiples of 32");
  USC_REQUIRE(split_cols >= 32 && split_cols % 32 == 0 && N % split_cols == 0 && x2_cols >= 0 && x2_cols % 32 == 0,
              "usc_linear_fwd_split: split_cols / x2_cols must be multiples of 32, split_cols a divisor of N");
  USC_REQUIRE(x && W && y, "usc_linear_fwd: null pointer");
  hipLaunchKernelGGL(usc::linear_fwd_kernel, dim3(N / 32, (M + 31) / 32), dim3(256), 0, usc::as_stream(s), x, x2, W, b,
                     (int)M, (int)N, (int)K, 0, y, (int)M, (int)x2_cols, (int)split_cols);
  USC_CHECK_LAUNCH("usc_linear_fwd");
  return USC_OK;
}

int usc_linear_bwd(const float* dy, const float* x, const float* W, int32_t M, int32_t N, int32_t K, float* dx, float* dW,
                   float* db, int32_t accumulate, usc_stream_t s) {
  return usc_linear_bwd_ex2(dy, nullptr, x, nullptr, W, M, N, K, dx, nullptr, nullptr, nullptr, dW, db, accumulate, s);
}

int usc_linear_bwd_ex(const float* dy, const float* y_relu, const float* x, const float* x2, const float* W, int32_t M,
                      int32_t N, int32_t K, float* dx, const float* dx_add, float* dW, float* db, int32_t accumulate,
                      usc_stream_t s) {
  return usc_linear_bwd_ex2(dy, y_relu, x, x2, W, M, N, K, dx, dx_add, nullptr, nullptr, dW, db, accumulate, s);
}

int usc_linear_bwd_ex2(const float* dy, const float* y_relu, const float* x, const float* x2, const float* W, int32_t M,
                       int32_t N, int32_t K, float* dx, const float* dx_add, const float* dx_add2, float* dx_b, float* dW,
                       float* db, int32_t accumulate, usc_stream_t s) {
  USC_REQUIRE(M >= 1 && N >= 32 && N % 32 == 0 && K >= 32 && K % 32 == 0, "usc_linear_bwd: N, K must be multiples of 32");
  USC_REQUIRE(dy && x && W, "usc_linear_bwd: null pointer");
  USC_REQUIRE(dx || !(dx_add || dx_add2 || dx_b), "usc_linear_bwd: dx_add / dx_add2 / dx_b need dx");
  hipStream_t st = usc::as_stream(s);
  const int dx_tiles = dx ? (K / 32) * ((M + 31) / 32) : 0;
  const int dw_tiles = dW ? (K / 32) * (N / 32) : 0;
  if (dx_tiles + dw_tiles > 0) {
    if (y_relu || x2 || dx_add || dx_add2 || dx_b)
      hipLaunchKernelGGL(usc::linear_bwd_kernel<true>, dim3(dx_tiles + dw_tiles), dim3(256), 0, st, dy, y_relu, x, x2, W,
                         (int)M, (int)N, (int)K, dx_tiles, (int)accumulate, dx, dx_add, dx_add2, dx_b, dW, db);
    else
      hipLaunchKernelGGL(usc::linear_bwd_kernel<false>, dim3(dx_tiles + dw_tiles), dim3(256), 0, st, dy, y_relu, x, x2, W,
                         (int)M, (int)N, (int)K, dx_tiles, (int)accumulate, dx, dx_add, dx_add2, dx_b, dW, db);
  }
  USC_CHECK_LAUNCH("usc_linear_bwd");
  return USC_OK;
}

int usc_qkv_proj_bwd(const float* dy3, const float* x, const float* pos, const float* W, int32_t M, int32_t E, float* dx,
                     float* dx_b, const float* dres, float* dW, float* db, int32_t accumulate, usc_stream_t s) {
  USC_REQUIRE(M >= 1 && E >= 32 && E % 32 == 0, "usc_qkv_proj_bwd: E must be a multiple of 32");
  USC_REQUIRE(dy3 && x && W && dx && dW, "usc_qkv_proj_bwd: null pointer");
  const int dx_tiles = (E / 32) * ((M + 31) / 32);
  const int dw_tiles = (E / 32) * (3 * E / 32);
  hipLaunchKernelGGL(usc::qkv_bwd_kernel, dim3(dx_tiles + dw_tiles), dim3(256), 0, usc::as_stream(s), dy3, x, pos, W, (int)M,
                     (int)E, dx_tiles, (int)accumulate, dx, dx_b, dres, dW, db);
  USC_CHECK_LAUNCH("usc_qkv_proj_bwd");
  return USC_OK;
}

int64_t usc_col_sum_ws_bytes(int64_t n, int32_t c) {
  const int64_t parts = n > 0 ? (n + usc::kColSumRows - 1) / usc::kColSumRows : 0;
  return parts * (c > 0 ? c : 0) * 4;
}

int usc_col_sum(const float* x, int64_t n, int32_t c, float* out, int32_t accumulate, void* ws, int64_t ws_bytes,
                usc_stream_t s) {
  USC_REQUIRE(n >= 0 && c >= 1, "usc_col_sum: bad sizes");
  USC_REQUIRE(out && (n == 0 || (x && ws)), "usc_col_sum: null pointer");
  USC_REQUIRE(ws_bytes >= usc_col_sum_ws_bytes(n, c), "usc_col_sum: workspace too small");
  hipStream_t st = usc::as_stream(s);
  const int64_t parts = n > 0 ? (n + usc::kColSumRows - 1) / usc::kColSumRows : 0;
  const unsigned cg = (unsigned)((c + 63) / 64);
  if (parts > 0)
    hipLaunchKernelGGL(usc::col_sum_partial_kernel, dim3((unsigned)parts, cg), dim3(256), 0, st, x, n, (int)c, (float*)ws);
  hipLaunchKernelGGL(usc::col_sum_final_kernel, dim3((unsigned)((c + 15) / 16)), dim3(256), 0, st, (const float*)ws, parts,
                     (int)c, (int)accumulate, out);
  USC_CHECK_LAUNCH("usc_col_sum");
  return USC_OK;
}

}  // extern "C"
