// Which clock does the chip hold under float64 VALU load?  (MI355X_MICROARCH.md, DVFS give-back item 6: the in-kernel
// clock is delta s_memtime / delta s_memrealtime x 100 MHz, stamped around the loop after >= 2 s of back-to-back launches.)
// The FP64-VALU peak bench.py prices against -- 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-slots/s -- is the
// data-sheet clock; a kernel that keeps the float64 pipes busy runs at the clock measured here.
//   KIND 0: dense v_fma_f64, eight independent chains per thread, 8 waves per SIMD (the issue-bound extreme)
//   KIND 1: one dependent chain of v_fma_f64 per thread, 2 waves per SIMD (latency-bound, the pipes mostly idle)
//   KIND 2: four chains per thread, 3 waves per SIMD (about the issue density of the bank kernels: ~65 % VALU-busy)
// build: hipcc --offload-arch=gfx950 -O3 -o f64_clock tools/microbench/f64_clock.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(256) k(double *out, unsigned long long *stamps, double a, double b, int iters) {
    constexpr int C = KIND == 0 ? 8 : KIND == 1 ? 1 : 4;
    double x[C];
#pragma unroll
    for (int i = 0; i < C; ++i) x[i] = a + threadIdx.x * 1e-9 + i;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8 / C; ++u)
#pragma unroll
            for (int i = 0; i < C; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < C; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2 + 0] = c1 - c0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

template <int KIND>
void run(const char *name, int wgs_per_cu, double *out, unsigned long long *stamps) {
    const int iters = 8192, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    float ms = 0;
    // >= 2 s of back-to-back launches, then one timed and stamped launch
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.2) {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, stamps, 1.0000001, 1e-7, iters);
        (void)hipDeviceSynchronize();
        launches += 50;
    }
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, stamps, 1.0000001, 1e-7, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 2);
    (void)hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> mhz;
    for (int i = 0; i < grid; ++i)
        if (h[i * 2 + 1]) mhz.push_back(100.0 * (double)h[i * 2] / (double)h[i * 2 + 1]);
    std::sort(mhz.begin(), mhz.end());
    const double lane_ops = (double)grid * 256 * iters * 8;
    const double clock = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
    printf("%-44s %7.3f ms/launch  %6.2f T lane-slots/s  in-kernel clock %6.0f MHz (min %.0f max %.0f)  "
           "%.2f cycles per wave instruction at that clock  [%d launches before]\n",
           name, ms, lane_ops / (ms * 1e-3) / 1e12, clock, mhz.empty() ? 0.0 : mhz.front(), mhz.empty() ? 0.0 : mhz.back(),
           (ms * 1e-3) * clock * 1e6 * 1024 / (lane_ops / 64), launches);
}

int main() {
    double *out;
    unsigned long long *stamps;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
    (void)hipMalloc(&stamps, 256 * 8 * 2 * sizeof(unsigned long long));
    run<0>("dense v_fma_f64 (8 chains, 8 waves/SIMD)", 8, out, stamps);
    run<1>("one dependent chain (2 waves/SIMD)", 2, out, stamps);
    run<2>("four chains (3 waves/SIMD)", 3, out, stamps);
    return 0;
}
