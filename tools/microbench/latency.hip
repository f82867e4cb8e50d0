// Calibration of the latency-bound kernels: cycles per dependent instruction for one wave on an otherwise
// idle MI355X, and the shader clock it actually runs at.  hipcc --offload-arch=gfx950 -O3 latency.hip -o latency
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ void chain(double *out, long long *ticks, int iters, double x0, double m, double c) {
    double x = x0 + threadIdx.x * 1e-9;
    __syncthreads();
    const long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) x = __builtin_fma(x, m, c);                       // dependent v_fma_f64
            if (KIND == 1) x = (x > c) ? x * m : x + c;                      // cmp + select + op
            if (KIND == 2) { float f = (float)x; f = __builtin_fmaf(f, (float)m, (float)c); x = f; }
            if (KIND == 3) {                                                 // dependent DPP move pair + fma
                const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x111, 0xf, 0xf, false);
                const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x111, 0xf, 0xf, false);
                x = __builtin_fma(__hiloint2double(hi, lo), m, c);
            }
            if (KIND == 4) { __shared__ double s[1024]; s[threadIdx.x] = x; __syncthreads(); x = s[threadIdx.x ^ 1] * m + c; }
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = c1 - c0; ticks[1] = w1 - w0; }
}

template <int KIND>
void run(const char *name, int threads, int blocks) {
    double *out; long long *ticks, h[2];
    hipMalloc(&out, sizeof(double) * threads * blocks);
    hipMalloc(&ticks, 16);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(chain<KIND>, dim3(blocks), dim3(threads), 0, 0, out, ticks, iters, 0.5, 0.999, 0.001);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
    const double n = iters * 16.0;
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);      // kHz
    const double ns = h[1] / (rate * 1e-6) / n;
    printf("%-34s threads %4d blocks %3d: %6.1f clk/iter  %6.2f ns/iter  (clock64 %.0f MHz)\n", name, threads, blocks,
           h[0] / n, ns, h[0] / (h[1] / (rate * 1e-3)) );
    hipFree(out); hipFree(ticks);
}

int main() {
    run<0>("dependent v_fma_f64", 64, 1);
    run<0>("dependent v_fma_f64", 256, 1);
    run<0>("dependent v_fma_f64", 512, 1);
    run<0>("dependent v_fma_f64", 1024, 1);
    run<0>("dependent v_fma_f64", 256, 1024);
    run<1>("f64 cmp + 2 ops + select", 64, 1);
    run<1>("f64 cmp + 2 ops + select", 512, 1);
    run<2>("cvt + v_fma_f32 + cvt", 64, 1);
    run<3>("2 dpp mov + v_fma_f64", 64, 1);
    run<3>("2 dpp mov + v_fma_f64", 512, 1);
    run<4>("lds write + barrier + read + 2 ops", 256, 1);
    run<4>("lds write + barrier + read + 2 ops", 512, 1);
    run<4>("lds write + barrier + read + 2 ops", 1024, 1);
    return 0;
}
