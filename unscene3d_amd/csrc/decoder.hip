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
template <int PER>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int64_t rows, int d, float eps,
                                                           float* __restrict__ y, float* __restrict__ mean,
                                                           float* __restrict__ rstd) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float v[PER];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < PER; ++j) { v[j] = x[r * d + lane + 64 * j]; s += v[j]; }
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
                                                            int64_t rows_per_block, float* __restrict__ dx,
                                                            float* __restrict__ part /* [G][2][d] or dgamma/dbeta */,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[16][2][64 * PER];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float g[PER], dg[PER], db[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) { g[j] = gamma[lane + 64 * j]; dg[j] = 0.f; db[j] = 0.f; }
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
    for (int j = 0; j < PER; ++j) dx[r * d + lane + 64 * j] = rs * (gy[j] - s1 - xh[j] * s2);
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) { red[wave][0][lane + 64 * j] = dg[j]; red[wave][1][lane + 64 * j] = db[j]; }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * d; c += 1024) {
    const int which = c / d, col = c - which * d;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w][which][col];
    if (gridDim.x == 1) (which == 0 ? dgamma : dbeta)[col] = s;
    else part[((int64_t)blockIdx.x * 2 + which) * d + col] = s;
  }
}

__global__ __launch_bounds__(256) void layernorm_bwd_final_kernel(const float* __restrict__ part, int G, int d,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= 2 * d) return;
  const int which = c / d, col = c - which * d;
  float s = 0.f;
  for (int b = 0; b < G; ++b) s += part[((int64_t)b * 2 + which) * d + col];
  (which == 0 ? dgamma : dbeta)[col] = s;
}

constexpr int64_t kLnRowsPerBlock = 1024;

}  // namespace
}  // namespace usc

using namespace usc;

extern "C" {

int usc_layernorm_fwd(const float* x, const float* gamma, const float* beta, int64_t rows, int32_t d, float eps, float* y,
                      float* mean, float* rstd, usc_stream_t s) {
  USC_REQUIRE(rows >= 0 && d >= 64 && d % 64 == 0 && d <= 64 * kLnMaxPerLane, "usc_layernorm_fwd: d must be a multiple of 64, <= 512");
  if (rows == 0) return USC_OK;
  USC_REQUIRE(x && gamma && beta && y && mean && rstd, "usc_layernorm_fwd: null pointer");
  const dim3 grid((unsigned)ceil_div(rows, 4));
  hipStream_t st = as_stream(s);
#define USC_LN_F(P) hipLaunchKernelGGL((layernorm_fwd_kernel<P>), grid, dim3(256), 0, st, x, gamma, beta, rows, (int)d, eps, y, mean, rstd)
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
                      int64_t rows, int32_t d, float* dx, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes,
                      usc_stream_t s) {
  USC_REQUIRE(rows >= 1 && d >= 64 && d % 64 == 0 && d <= 64 * kLnMaxPerLane, "usc_layernorm_bwd: bad sizes");
  USC_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta, "usc_layernorm_bwd: null pointer");
  const int64_t G = ceil_div(rows, kLnRowsPerBlock);
  USC_REQUIRE(G == 1 || (ws && ws_bytes >= G * 2 * d * 4), "usc_layernorm_bwd: workspace too small");
  hipStream_t st = as_stream(s);
#define USC_LN_B(P) hipLaunchKernelGGL((layernorm_bwd_kernel<P>), dim3((unsigned)G), dim3(1024), 0, st, dy, x, mean, rstd, gamma, rows, (int)d, kLnRowsPerBlock, dx, (float*)ws, dgamma, dbeta)
  switch (d / 64) {
    case 1: USC_LN_B(1); break;  case 2: USC_LN_B(2); break;  case 3: USC_LN_B(3); break;  case 4: USC_LN_B(4); break;
    case 6: USC_LN_B(6); break;  case 8: USC_LN_B(8); break;
    default: USC_REQUIRE(false, "usc_layernorm_bwd: unsupported d (64*{1,2,3,4,6,8})");
  }
#undef USC_LN_B
  if (G > 1)
    hipLaunchKernelGGL(layernorm_bwd_final_kernel, dim3((unsigned)ceil_div(2 * d, 256)), dim3(256), 0, st,
                       (const float*)ws, (int)G, (int)d, dgamma, dbeta);
  USC_CHECK_LAUNCH("usc_layernorm_bwd");
  return USC_OK;
}

}  // extern "C"
