// What does a wave get out of the float64 pipe on this part?  Independent chains of one instruction kind per
// thread (ILP = 8), 8 waves per SIMD: v_fma_f64, v_mul_f64 + v_add_f64 pairs, v_rndne_f64, v_rcp_f64, v_cvt.
// build: hipcc --offload-arch=gfx950 -O3 -o f64_rate tools/microbench/f64_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ void __launch_bounds__(256) k(double *out, double a, double b, int iters) {
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) x[i] = __builtin_fma(x[i], a, b);
            if (KIND == 1) x[i] = x[i] * a;
            if (KIND == 2) x[i] = x[i] + b;
            if (KIND == 3) x[i] = __builtin_rint(x[i]) + b;
            if (KIND == 4) x[i] = __builtin_amdgcn_rcp(x[i]);
            if (KIND == 5) x[i] = (double)(float)x[i] + b;
            if (KIND == 6) x[i] = __builtin_floor(x[i]) + b;
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, double ops_per_iter_per_lane, double *out) {
    const int iters = 4096, grid = 256 * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, 1.0000001, 1e-7, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, 1.0000001, 1e-7, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = (double)grid * 256 * iters * ops_per_iter_per_lane;
    printf("%-28s %8.3f ms  %7.2f T lane-ops/s  (%.2f cycles per wave instruction at 2.4 GHz, 1024 SIMDs)\n", name, ms,
           lane_ops / (ms * 1e-3) / 1e12, (ms * 1e-3) * 2.4e9 * 1024 / (lane_ops / 64));
}

int main() {
    double *out;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
    run<0>("v_fma_f64", 8, out);
    run<1>("v_mul_f64", 8, out);
    run<2>("v_add_f64", 8, out);
    run<3>("v_rndne_f64 + v_add_f64", 16, out);
    run<4>("v_rcp_f64", 8, out);
    run<5>("cvt f64->f32->f64 + add", 24, out);
    run<6>("v_floor_f64 + v_add_f64", 16, out);
    return 0;
}
