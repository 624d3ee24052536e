// points.hip — point-set kernels of the Mask3D query path (SURVEY.md §8a rows
// Q1, Q2): furthest point sampling with the reference kernel's exact tie-break,
// and the Fourier positional encoding.
#include "common.h"

namespace usc {

// Candidate ordering of the reference kernel (sampling_gpu.cu:62-70, :100-116):
// larger distance wins; among equal distances the reference's strided per-thread
// scan + tree reduction keeps the candidate with the smallest (k mod BS), then
// the smallest k, where BS = opt_n_threads(n) is the reference block size.
struct FpsCand {
  float d;
  int k;
};
__device__ inline bool fps_better(const FpsCand& a, const FpsCand& b, int bs_mask) {
  if (a.d > b.d) return true;
  if (a.d < b.d) return false;
  const int ra = a.k & bs_mask, rb = b.k & bs_mask;
  if (ra != rb) return ra < rb;
  return a.k < b.k;
}

constexpr int kFpsThreads = 1024;

__global__ __launch_bounds__(kFpsThreads) void fps_kernel(const float* __restrict__ xyz, int n, int m, int bs_mask,
                                                         float* __restrict__ tmp, int32_t* __restrict__ idx) {
  if (m <= 0) return;
  __shared__ float sd[kFpsThreads / 64];
  __shared__ int sk[kFpsThreads / 64];
  __shared__ int s_old;
  const int b = blockIdx.x;
  xyz += (int64_t)b * n * 3;
  tmp += (int64_t)b * n;
  idx += (int64_t)b * m;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int old = 0;
  if (tid == 0) idx[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
    FpsCand best{-1.f, 0};
    for (int k = tid; k < n; k += kFpsThreads) {
      const float x2 = xyz[k * 3 + 0], y2 = xyz[k * 3 + 1], z2 = xyz[k * 3 + 2];
      const float mag = (x2 * x2) + (y2 * y2) + (z2 * z2);
      if (mag <= 1e-3f) continue;
      const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
      const float d2 = fminf(d, tmp[k]);
      tmp[k] = d2;
      FpsCand c{d2, k};
      if (fps_better(c, best, bs_mask)) best = c;
    }
    // wave argmax
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      FpsCand other{__shfl_xor(best.d, o, 64), __shfl_xor(best.k, o, 64)};
      if (fps_better(other, best, bs_mask)) best = other;
    }
    if (lane == 0) {
      sd[wave] = best.d;
      sk[wave] = best.k;
    }
    __syncthreads();
    if (wave == 0) {
      FpsCand c{-1.f, 0};
      if (lane < kFpsThreads / 64) c = FpsCand{sd[lane], sk[lane]};
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        FpsCand other{__shfl_xor(c.d, o, 64), __shfl_xor(c.k, o, 64)};
        if (fps_better(other, c, bs_mask)) c = other;
      }
      if (lane == 0) {
        s_old = c.k;
        idx[j] = c.k;
      }
    }
    __syncthreads();
    old = s_old;
  }
}

// ---------------------------------------------------------------------------
// Multi-workgroup FPS.  The reference kernel is one block per scene that re-reads all n points
// from memory for each of the m samples.  On MI355X the point set is spread over G workgroups
// (G <= 240, all co-resident) that keep their points AND their running min-distances in registers;
// each iteration every workgroup publishes its best candidate with one 64-bit atomicMax and the
// workgroups meet at a monotonic arrival counter.  The candidate is packed so that integer max ==
// the reference ordering (distance, then smallest k mod BS, then smallest k); 0 = "no candidate".
//   hand-off protocol (cdna_hip_programming.md G16): relaxed agent-scope atomicMax on best[j] ->
//   release agent-scope add on the counter; consumers poll relaxed, then acquire, then read best[j]
//   with an agent-scope atomic load.  best[]/counter live in caller scratch zeroed before launch.
constexpr int kFpsMaxSlots = 12;       // points per thread held in registers
constexpr int kFpsGroupThreads = 256;

__device__ inline unsigned long long fps_pack(float d, int k, int bs_mask) {
  const unsigned int tie = ((unsigned int)(k & bs_mask) << 22) | (unsigned int)k;   // k < 2^22
  return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(0x7fffffffu - tie);
}

__global__ __launch_bounds__(kFpsGroupThreads) void fps_multi_kernel(const float* __restrict__ xyz, int n, int m,
                                                                    int bs_mask, int G,
                                                                    unsigned long long* __restrict__ scratch,
                                                                    int32_t* __restrict__ idx) {
  __shared__ unsigned long long s_best[kFpsGroupThreads / 64];
  __shared__ int s_old;
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  xyz += (int64_t)b * n * 3;
  idx += (int64_t)b * m;
  unsigned long long* best = scratch + (int64_t)b * (m + 2);     // best[0..m) | counter
  unsigned int* counter = reinterpret_cast<unsigned int*>(best + m);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int stride = G * kFpsGroupThreads;

  float px[kFpsMaxSlots], py[kFpsMaxSlots], pz[kFpsMaxSlots], td[kFpsMaxSlots];
  bool ok[kFpsMaxSlots];
#pragma unroll
  for (int q = 0; q < kFpsMaxSlots; ++q) {
    const int k = g * kFpsGroupThreads + tid + q * stride;
    const bool in = k < n;
    px[q] = in ? xyz[k * 3 + 0] : 0.f;
    py[q] = in ? xyz[k * 3 + 1] : 0.f;
    pz[q] = in ? xyz[k * 3 + 2] : 0.f;
    const float mag = (px[q] * px[q]) + (py[q] * py[q]) + (pz[q] * pz[q]);
    ok[q] = in && !(mag <= 1e-3f);
    td[q] = 1e10f;
  }
  int old = 0;
  if (g == 0 && tid == 0) idx[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
    unsigned long long v = 0ull;
#pragma unroll
    for (int q = 0; q < kFpsMaxSlots; ++q) {
      if (ok[q]) {
        const float d = (px[q] - x1) * (px[q] - x1) + (py[q] - y1) * (py[q] - y1) + (pz[q] - z1) * (pz[q] - z1);
        const float d2 = fminf(d, td[q]);
        td[q] = d2;
        const unsigned long long c = fps_pack(d2, g * kFpsGroupThreads + tid + q * stride, bs_mask);
        v = c > v ? c : v;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(v, o, 64);
      v = other > v ? other : v;
    }
    if (lane == 0) s_best[wave] = v;
    __syncthreads();
    if (tid == 0) {
      unsigned long long w = s_best[0];
#pragma unroll
      for (int q = 1; q < kFpsGroupThreads / 64; ++q) w = s_best[q] > w ? s_best[q] : w;
      if (w) __hip_atomic_fetch_max(&best[j], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int target = (unsigned int)G * (unsigned int)j;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
        __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      const unsigned long long win = __hip_atomic_load(&best[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int k = 0;
      if (win) k = (int)((0x7fffffffu - (unsigned int)(win & 0xffffffffull)) & 0x3fffffu);
      s_old = k;
      if (g == 0) idx[j] = k;
    }
    __syncthreads();
    old = s_old;
  }
}

__global__ __launch_bounds__(256) void fourier_posenc_kernel(const float* __restrict__ xyz, int64_t n,
                                                            const float* __restrict__ lo,
                                                            const float* __restrict__ hi,
                                                            const float* __restrict__ B, int d,
                                                            float* __restrict__ out) {
  const int half = d >> 1;
  const int64_t total = n * half;
  const float two_pi = 6.283185307179586f;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = j / half;
    const int f = (int)(j - r * half);
    float proj = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float l = lo[a], h = hi[a];
      // shift_scale_points with dst_range [0,1]: ((x - lo) * 1) / (hi - lo) + 0
      const float xn = ((xyz[r * 3 + a] - l) * 1.0f) / (h - l) + 0.0f;
      proj = fmaf(xn * two_pi, B[a * half + f], proj);
    }
    out[r * d + f] = sinf(proj);
    out[r * d + half + f] = cosf(proj);
  }
}

}  // namespace usc

using namespace usc;

extern "C" {

int usc_furthest_point_sampling(const float* xyz, int32_t b, int32_t n, int32_t m, float* tmp, int32_t* idx,
                                usc_stream_t s) {
  USC_REQUIRE(b >= 0 && n >= 1 && m >= 0, "usc_furthest_point_sampling: bad sizes");
  if (b == 0 || m == 0) return USC_OK;
  USC_REQUIRE(xyz && tmp && idx, "usc_furthest_point_sampling: null pointer");
  // reference block size: opt_n_threads(n) = min(2^floor(log2 n), 512)  (cuda_utils.h:17-21)
  int bs = 1;
  while (bs * 2 <= n && bs < 512) bs *= 2;
  hipStream_t st = as_stream(s);
  // workgroups per scene for the register-resident kernel
  int G = (int)ceil_div(n, (int64_t)kFpsGroupThreads * 10);
  if (G < 1) G = 1;
  const bool fits = (int64_t)G * kFpsGroupThreads * kFpsMaxSlots >= n && n < (1 << 22) && (int64_t)b * G <= 240 &&
                    (int64_t)n * 4 >= (int64_t)(m + 2) * 8 && n >= 4096;
  if (fits) {
    // scratch (best[m] + arrival counter per scene, compact at the start of `tmp`) is carved from the
    // caller's buffer: the register-resident kernel keeps the running distances itself
    (void)hipMemsetAsync(tmp, 0, (size_t)b * (m + 2) * 8, st);
    hipLaunchKernelGGL(fps_multi_kernel, dim3((unsigned)(b * G)), dim3(kFpsGroupThreads), 0, st, xyz, (int)n, (int)m,
                       bs - 1, G, reinterpret_cast<unsigned long long*>(tmp), idx);
  } else {
    hipLaunchKernelGGL(fps_kernel, dim3((unsigned)b), dim3(kFpsThreads), 0, st, xyz, (int)n, (int)m, bs - 1, tmp, idx);
  }
  USC_CHECK_LAUNCH("usc_furthest_point_sampling");
  return USC_OK;
}

int usc_fourier_posenc(const float* xyz, int64_t n, const float* lo, const float* hi, const float* gauss_B, int32_t d,
                       float* out, usc_stream_t s) {
  USC_REQUIRE(n >= 0 && d >= 2 && d % 2 == 0, "usc_fourier_posenc: bad sizes");
  if (n == 0) return USC_OK;
  USC_REQUIRE(xyz && lo && hi && gauss_B && out, "usc_fourier_posenc: null pointer");
  hipLaunchKernelGGL(fourier_posenc_kernel, dim3(stream_grid(n * (d / 2), 256)), dim3(256), 0, as_stream(s), xyz, n, lo,
                     hi, gauss_B, (int)d, out);
  USC_CHECK_LAUNCH("usc_fourier_posenc");
  return USC_OK;
}

}  // extern "C"
