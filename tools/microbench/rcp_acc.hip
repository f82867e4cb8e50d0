// How accurate is v_rcp_f64 (and one / two Newton steps on it)?  Max relative error against 1/x over 2^24 arguments.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double *err, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long s = 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
    s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
    const double x = (1.0 + (double)(s >> 11) * 0x1p-53) * ((i & 1) ? 3.7e-5 : 91.3);
    const double t = 1.0 / x;
    double y0 = __builtin_amdgcn_rcp(x);
    double y1 = __builtin_fma(__builtin_fma(-x, y0, 1.0), y0, y0);
    double y2 = __builtin_fma(__builtin_fma(-x, y1, 1.0), y1, y1);
    err[i * 3 + 0] = fabs(y0 - t) / t;
    err[i * 3 + 1] = fabs(y1 - t) / t;
    err[i * 3 + 2] = fabs(y2 - t) / t;
}
int main() {
    const int n = 1 << 24;
    double *d; (void)hipMalloc(&d, (size_t)n * 3 * sizeof(double));
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, n);
    double *h = (double *)malloc((size_t)n * 3 * sizeof(double));
    (void)hipMemcpy(h, d, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost);
    double m[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) for (int k2 = 0; k2 < 3; ++k2) if (h[i * 3 + k2] > m[k2]) m[k2] = h[i * 3 + k2];
    printf("max relative error: v_rcp_f64 %.3e (2^%.1f)   + one Newton step %.3e   + two %.3e   (2^-53 = %.3e)\n", m[0],
           log2(m[0]), m[1], m[2], 0x1p-53);
    return 0;
}
