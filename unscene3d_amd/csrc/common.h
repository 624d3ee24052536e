// common.h — shared helpers for the gfx950 kernels of libusc3d_hip.so.
// CDNA4 only: wave = 64 lanes, no compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/usc3d.h"

namespace usc {

constexpr int kWave = 64;

// ---- error reporting -------------------------------------------------------
void set_error(const char* fmt, ...);

#define USC_REQUIRE(cond, ...)                   \
  do {                                           \
    if (!(cond)) {                               \
      ::usc::set_error(__VA_ARGS__);             \
      return USC_ERR_ARG;                        \
    }                                            \
  } while (0)

#define USC_CHECK_LAUNCH(name)                                                  \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      ::usc::set_error("%s: HIP launch failed: %s", name, hipGetErrorString(e__)); \
      return USC_ERR_LAUNCH;                                                    \
    }                                                                           \
  } while (0)

static inline hipStream_t as_stream(usc_stream_t s) { return (hipStream_t)s; }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t align_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// Grid for HBM-bound grid-stride kernels: enough blocks to fill 256 CUs x 8.
static inline int stream_grid(int64_t work_items, int block) {
  int64_t g = ceil_div(work_items, block);
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- coordinate key packing -----------------------------------------------
// (b, x, y, z) -> 64-bit key: b 10 bits | x 18 | y 18 | z 18, biased by 2^17.
constexpr int kCoordBits = 18;
constexpr int kCoordBias = 1 << (kCoordBits - 1);
constexpr uint64_t kEmptyKey = ~0ull;

__host__ __device__ inline bool coord_in_range(int b, int x, int y, int z) {
  return b >= 0 && b < 1024 && x >= -kCoordBias && x < kCoordBias &&
         y >= -kCoordBias && y < kCoordBias && z >= -kCoordBias && z < kCoordBias;
}

__host__ __device__ inline uint64_t pack_key(int b, int x, int y, int z) {
  return ((uint64_t)(uint32_t)b << (3 * kCoordBits)) |
         ((uint64_t)(uint32_t)(x + kCoordBias) << (2 * kCoordBits)) |
         ((uint64_t)(uint32_t)(y + kCoordBias) << kCoordBits) |
         (uint64_t)(uint32_t)(z + kCoordBias);
}

// 64-bit mix (splitmix64 finaliser) -> slot
__host__ __device__ inline uint64_t hash_key(uint64_t k) {
  k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull;
  k ^= k >> 27; k *= 0x94d049bb133111ebull;
  k ^= k >> 31;
  return k;
}

// floor division toward -inf for quantisation to multiples of q (q > 0)
__host__ __device__ inline int floor_quant(int c, int q) {
  int d = c / q;
  if ((c % q != 0) && ((c < 0) != (q < 0))) --d;
  return d * q;
}

#if defined(__HIPCC__)
// Lookup in an open-addressing table; returns row or -1.
__device__ inline int table_lookup(const uint64_t* __restrict__ keys,
                                   const int32_t* __restrict__ vals, int64_t cap,
                                   uint64_t key) {
  uint64_t slot = hash_key(key) & (uint64_t)(cap - 1);
  for (;;) {
    uint64_t k = keys[slot];
    if (k == key) return vals[slot];
    if (k == kEmptyKey) return -1;
    slot = (slot + 1) & (uint64_t)(cap - 1);
  }
}

// wave-level inclusive/exclusive helpers (64 lanes)
__device__ inline int lane_id() { return threadIdx.x & 63; }

__device__ inline int wave_reduce_add(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_reduce_addf(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline double wave_reduce_addd(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// inclusive prefix sum over the 64 lanes of a wave
__device__ inline int wave_inclusive_scan(int v) {
  const int l = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(v, o, 64);
    if (l >= o) v += t;
  }
  return v;
}
#endif

}  // namespace usc
