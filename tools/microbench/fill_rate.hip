// What does a write-only kernel need to reach hipMemsetAsync's rate on this part (6.5 TB/s at 536 MB; the library's
// grid-stride fill: 4.7)?  Variants of a float fill over a 536 MB buffer (beyond the memory-side cache), HIP events over
// 100 launches after 30.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/fill_rate tools/microbench/fill_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

// A: grid-stride loop, one 16-byte store per thread and trip (the library's k_fill)
__global__ void __launch_bounds__(256) fill_a(float *out, long n4, float v) {
    const long stride = (long)gridDim.x * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += stride) reinterpret_cast<v4f *>(out)[e] = v4f{v, v, v, v};
}
// B: one workgroup per contiguous chunk of U x 4 KB, no loop over the grid
template <int U, bool NT>
__global__ void __launch_bounds__(256) fill_b(float *out, long n4, float v) {
    const long base = (long)blockIdx.x * 256 * U + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long e = base + (long)u * 256;
        if (e < n4) {
            if (NT) __builtin_nontemporal_store(v4f{v, v, v, v}, reinterpret_cast<v4f *>(out) + e);
            else reinterpret_cast<v4f *>(out)[e] = v4f{v, v, v, v};
        }
    }
}
// C: each wave owns a contiguous run (a tile of the filter kernels: 64 lanes x 16 floats = 4 KB, four 1 KB stores)
__global__ void __launch_bounds__(256) fill_c(float *out, long n4, float v, int tiles_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = 0; t < tiles_per_wg; ++t) {
        const long tile = (long)blockIdx.x * tiles_per_wg + t;          // 4096 floats per workgroup-tile
        float *dst = out + tile * 4096 + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long e = (tile * 4096 + wave * 1024 + i * 256 + lane * 4) / 4;
            if (e < n4) *reinterpret_cast<v4f *>(dst + i * 256 + lane * 4) = v4f{v, v, v, v};
        }
    }
}

template <typename F>
static void timeit(const char *name, long bytes, F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 30; ++i) launch();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 100; ++i) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %8.2f us  %5.2f TB/s\n", name, ms * 10.0, bytes / (ms * 1e-5) / 1e12);
}

int main() {
    const long n = 134000000, n4 = n / 4, bytes = n * 4;
    float *out;
    (void)hipMalloc(&out, bytes);
    timeit("hipMemsetAsync", bytes, [&] { (void)hipMemsetAsync(out, 0, bytes, 0); });
    for (int wgs : {2048, 4096, 8192, 16384})
        timeit((std::string("A grid-stride, 16 B per trip, ") + std::to_string(wgs) + " workgroups").c_str(), bytes,
               [&] { hipLaunchKernelGGL(fill_a, dim3(wgs), dim3(256), 0, 0, out, n4, 0.25f); });
    timeit("B one chunk per workgroup, 4 x 16 B per thread", bytes,
           [&] { hipLaunchKernelGGL((fill_b<4, false>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, out, n4, 0.25f); });
    timeit("B ... 16 x 16 B per thread", bytes,
           [&] { hipLaunchKernelGGL((fill_b<16, false>), dim3((unsigned)((n4 + 4095) / 4096)), dim3(256), 0, 0, out, n4, 0.25f); });
    timeit("B ... 16 x 16 B per thread, non-temporal", bytes,
           [&] { hipLaunchKernelGGL((fill_b<16, true>), dim3((unsigned)((n4 + 4095) / 4096)), dim3(256), 0, 0, out, n4, 0.25f); });
    timeit("B ... 4 x 16 B per thread, non-temporal", bytes,
           [&] { hipLaunchKernelGGL((fill_b<4, true>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, out, n4, 0.25f); });
    for (int tpw : {1, 8, 43})
        timeit((std::string("C filter-kernel layout (wave-contiguous 4 KB), ") + std::to_string(tpw) + " tiles per workgroup").c_str(), bytes,
               [&] { hipLaunchKernelGGL(fill_c, dim3((unsigned)((n / 4096 + tpw - 1) / tpw)), dim3(256), 0, 0, out, n4, 0.25f, tpw); });
    return 0;
}
