// Do kernels of two HIP streams run at the same time on this box?  N spin kernels per stream (each `us` microseconds,
// `wgs` workgroups of 256 lanes): wall time for one stream alone, two streams, and two streams with the second one
// created at high priority / non-blocking.  Build: hipcc --offload-arch=gfx950 -O2 tools/probes/queue_overlap.hip -o
// tools/probes/queue_overlap ; run on the GPU box: tools/probes/queue_overlap [us=20] [wgs=1] [n=300]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void spin_kernel(long ticks, int* sink) {
    long t0 = wall_clock64();                       // 100 MHz constant clock
    while (wall_clock64() - t0 < ticks) {}
    if (ticks < 0) *sink = 1;
}

static double run(hipStream_t a, hipStream_t b, int n, long ticks, int wgs, int* sink) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, a, ticks, sink);
        if (b) hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, b, ticks, sink);
    }
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char** argv) {
    int us = argc > 1 ? atoi(argv[1]) : 20, wgs = argc > 2 ? atoi(argv[2]) : 1, n = argc > 3 ? atoi(argv[3]) : 300;
    long ticks = 100L * us;
    int* sink; hipMalloc(&sink, 4);
    hipStream_t s0, s1, s2, s3;
    hipStreamCreate(&s0); hipStreamCreate(&s1);
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi);
    hipStreamCreateWithFlags(&s3, hipStreamNonBlocking);
    run(s0, s1, 20, ticks, wgs, sink);
    double one = run(s0, nullptr, n, ticks, wgs, sink);
    double two = run(s0, s1, n, ticks, wgs, sink);
    double pri = run(s0, s2, n, ticks, wgs, sink);
    double nb = run(s0, s3, n, ticks, wgs, sink);
    double nul = run(nullptr, s1, n, ticks, wgs, sink);
    double nnb = run(nullptr, s3, n, ticks, wgs, sink);
    printf("spin %d us x %d launches, %d workgroups (priority range %d..%d)\n", us, n, wgs, lo, hi);
    printf("  one stream                      %8.1f us  (%.2f us per launch)\n", one, one / n);
    printf("  two streams                     %8.1f us  (x%.2f)\n", two, two / one);
    printf("  + high-priority second stream   %8.1f us  (x%.2f)\n", pri, pri / one);
    printf("  + non-blocking second stream    %8.1f us  (x%.2f)\n", nb, nb / one);
    printf("  null stream + created stream    %8.1f us  (x%.2f)\n", nul, nul / one);
    printf("  null stream + non-blocking      %8.1f us  (x%.2f)\n", nnb, nnb / one);
    return 0;
}
