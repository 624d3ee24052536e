// What rate does v_mfma_f32_32x32x2_f32 really sustain on gfx950?  Pure register loops, no memory: W waves per SIMD, each
// cycling over NACC independent accumulators (the dependent distance of the conv kernels is NACC issue slots).  The
// roofline fractions in bench.py are priced against 157.3 TFLOP/s (256 CUs x 4 SIMDs x 64 flop/cycle x 2.4 GHz); this
// probe says how much of that a loop with nothing else in it reaches, and at which clock.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_rate tools/probes/mfma_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int n = 0; n < NACC; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-9f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < NACC; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int NACC>
static void run(int waves_per_simd, int iters) {
  float* out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int cus = 256;
  // 256-thread workgroups = 4 waves = one per SIMD; waves_per_simd workgroups per CU
  dim3 grid(cus * waves_per_simd), block(256);
  hipLaunchKernelGGL(mfma_loop<NACC>, grid, block, 0, 0, out, 16, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<NACC>, grid, block, 0, 0, out, iters, 1.f, 1.f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid.x * 4 /*waves*/ * iters * 4.0 * NACC * 4096.0;
  printf("NACC %d, %d waves/SIMD: %.3f ms, %.1f TFLOP/s (%.3f of 157.3)\n", NACC, waves_per_simd, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
  hipFree(out);
}

int main() {
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("device clock attribute: %d kHz\n", clk);
  for (int w : {1, 2, 4}) {
    run<1>(w, 20000 / w);
    run<2>(w, 10000 / w);
    run<3>(w, 8000 / w);
    run<4>(w, 6000 / w);
  }
  // a long run: does the clock sag under sustained matrix-core load?
  run<3>(4, 60000);
  return 0;
}
