// scan.h — device-wide exclusive prefix sum of functor-generated int32 values.
// Three launches (block partials -> single-block scan of partials -> final),
// built on 64-lane wave shuffles.  Used for first-occurrence compaction of
// coordinate maps and for rulebook compaction.
#pragma once
#include "common.h"

namespace usc {

constexpr int kScanThreads = 256;
constexpr int kScanItemsPerThread = 8;
constexpr int kScanTile = kScanThreads * kScanItemsPerThread;  // 2048

static inline int64_t scan_num_blocks(int64_t n) { return n > 0 ? ceil_div(n, kScanTile) : 1; }
// scratch: one int64 per block (+1 for the grand total)
static inline int64_t scan_ws_bytes(int64_t n) { return (scan_num_blocks(n) + 1) * (int64_t)sizeof(int64_t); }

#if defined(__HIPCC__)
// Exclusive scan of one int per thread across a 256-thread block.
// Returns the exclusive prefix; *total gets the block sum (all threads).
__device__ inline int block_exclusive_scan_256(int v, int* total) {
  __shared__ int wave_sums[kScanThreads / 64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = wave_inclusive_scan(v);
  if (l == 63) wave_sums[w] = inc;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kScanThreads / 64; ++i) {
    int s = wave_sums[i];
    if (i < w) off += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return off + inc - v;
}

template <class F>
__global__ __launch_bounds__(kScanThreads) void scan_partials_kernel(F f, int64_t n, int64_t* block_sums) {
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItemsPerThread;
  int s = 0;
#pragma unroll
  for (int j = 0; j < kScanItemsPerThread; ++j) {
    int64_t i = base + j;
    if (i < n) s += f(i);
  }
  int tot;
  (void)block_exclusive_scan_256(s, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// In-place exclusive scan of block_sums[0..nb), grand total -> block_sums[nb].
static __global__ __launch_bounds__(kScanThreads) void scan_block_sums_kernel(int64_t* block_sums, int64_t nb) {
  __shared__ long long carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t c0 = 0; c0 < nb; c0 += kScanThreads) {
    int64_t i = c0 + threadIdx.x;
    long long v = (i < nb) ? block_sums[i] : 0;
    // 64-bit block scan via two-level wave scan
    __shared__ long long ws[kScanThreads / 64];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    long long inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      long long t = __shfl_up(inc, o, 64);
      if (l >= o) inc += t;
    }
    if (l == 63) ws[w] = inc;
    __syncthreads();
    long long off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < kScanThreads / 64; ++k) {
      long long sv = ws[k];
      if (k < w) off += sv;
      tot += sv;
    }
    long long carry = carry_s;
    if (i < nb) block_sums[i] = carry + off + inc - v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) block_sums[nb] = carry_s;
}

// out(i, exclusive_prefix, value) is invoked for every i < n.
template <class F, class O>
__global__ __launch_bounds__(kScanThreads) void scan_final_kernel(F f, O out, int64_t n, const int64_t* block_offsets) {
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItemsPerThread;
  int v[kScanItemsPerThread];
  int s = 0;
#pragma unroll
  for (int j = 0; j < kScanItemsPerThread; ++j) {
    int64_t i = base + j;
    v[j] = (i < n) ? f(i) : 0;
    s += v[j];
  }
  int tot;
  int ex = block_exclusive_scan_256(s, &tot);
  int64_t run = block_offsets[blockIdx.x] + ex;
#pragma unroll
  for (int j = 0; j < kScanItemsPerThread; ++j) {
    int64_t i = base + j;
    if (i < n) out(i, run, v[j]);
    run += v[j];
  }
}

// Host driver.  ws must hold scan_ws_bytes(n).  After completion ws[nb] (int64)
// holds the grand total (device memory).
template <class F, class O>
inline int device_exclusive_scan(F f, O out, int64_t n, void* ws, hipStream_t st) {
  int64_t nb = scan_num_blocks(n);
  int64_t* bs = (int64_t*)ws;
  hipLaunchKernelGGL(scan_partials_kernel<F>, dim3((unsigned)nb), dim3(kScanThreads), 0, st, f, n, bs);
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(kScanThreads), 0, st, bs, nb);
  hipLaunchKernelGGL((scan_final_kernel<F, O>), dim3((unsigned)nb), dim3(kScanThreads), 0, st, f, out, n, bs);
  return 0;
}
#endif

}  // namespace usc
