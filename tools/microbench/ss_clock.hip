// Which clock does the chip hold while k_supersaw_wide<4> -- the library's own kernel, included -- renders the 512-instance
// SuperSaw bank?  (MI355X_MICROARCH.md, DVFS give-back item 6: delta s_memtime / delta s_memrealtime x 100 MHz around the
// kernel's body, stamped after >= 2 s of back-to-back launches.)  tools/microbench/f64_clock.hip asks the same of plain
// v_fma_f64 loops; this one of the kernel the FP64 roofline of the SuperSaw mix is about.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I include -I pygmu2_amd/csrc tools/microbench/ss_clock.hip -o tools/microbench/ss_clock
#include <hip/hip_runtime.h>
__device__ unsigned long long g_stamps[2048][4];
#define PGX_SS_STAMP(i)                                                                             \
    do {                                                                                            \
        if (threadIdx.x == 0 && ((i) == 0 || (i) == 15)) {                                          \
            unsigned long long *s_ = g_stamps[blockIdx.y * gridDim.x + blockIdx.x] + ((i) == 0 ? 0 : 2); \
            s_[0] = __builtin_amdgcn_s_memtime();                                                   \
            s_[1] = __builtin_amdgcn_s_memrealtime();                                               \
        }                                                                                           \
    } while (0)
#include "../../pygmu2_amd/csrc/pgx_scan.hip"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <vector>

namespace pgx {
static thread_local std::string g_err;
void set_error(const std::string &m) { g_err = m; }
int fail(int code, const std::string &m) { g_err = m; return code; }
hipStream_t stream() { return nullptr; }
hipStream_t main_stream() { return nullptr; }
bool initialised() { return true; }
int device_index() { return 0; }
}  // namespace pgx
extern "C" int pgx_memset(void *, int, size_t) { return 0; }      // (referenced by an entry point this tool never calls)

int main(int argc, char **argv) {
    const int batch = argc > 1 ? atoi(argv[1]) : 512, nv = 7;
    const int64_t n = argc > 2 ? atoll(argv[2]) : 48000;
    std::vector<pgx_blitsaw_params> hp(batch * nv);
    std::vector<double> hs(batch * nv * 2), ha(batch, 0.3);
    for (int i = 0; i < batch; ++i)
        for (int v = 0; v < nv; ++v) {
            const double f = 55.0 * std::pow(2.0, (i * (512 / batch)) / 96.0) * std::pow(2.0, (v - 3) * 20.0 / 3.0 / 1200.0);
            hp[i * nv + v] = pgx_blitsaw_params{f, 1.0 / nv, 0.999, -1.0};
            hs[(i * nv + v) * 2] = std::fmod(0.37 * (i * nv + v), 1.0);
            hs[(i * nv + v) * 2 + 1] = 0.0;
        }
    pgx_blitsaw_params *dp;
    double *ds, *ds2, *da, *dtw;
    float *out;
    (void)hipMalloc(&dp, hp.size() * sizeof(hp[0]));
    (void)hipMalloc(&ds, hs.size() * 8);
    (void)hipMalloc(&ds2, hs.size() * 8);
    (void)hipMalloc(&da, ha.size() * 8);
    const int ring = argc > 3 ? atoi(argv[3]) : 1;              // output buffers used in turn (1: the same 98 MB every launch --
    std::vector<float *> outs(ring);                            // it then lives in the memory-side cache; 4: every launch writes to HBM)
    for (int r = 0; r < ring; ++r) (void)hipMalloc(&outs[r], (size_t)batch * n * 4);
    out = outs[0];
    int turn = 0;
    (void)hipMalloc(&dtw, (size_t)batch * nv * kSswTabDoubles * 8);
    (void)hipMemcpy(dp, hp.data(), hp.size() * sizeof(hp[0]), hipMemcpyHostToDevice);
    (void)hipMemcpy(ds, hs.data(), hs.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(da, ha.data(), ha.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_supersaw_wide_tables, dim3(batch), dim3(64), 0, 0, dtw, nv, 48000.0, dp);
    const size_t lds = (size_t)nv * (256 * 32 + kSswTabDoubles * 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_supersaw_wide<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int whole = 1 << 20;
    auto launch = [&]() {
        out = outs[turn++ % ring];
        hipLaunchKernelGGL(k_supersaw_wide<4>, dim3(batch, 1), dim3(256), lds, 0, out, n, nv, n, 1, (const double *)ds, ds2,
                           (const double *)da, whole, (const double *)dtw, 1);
    };
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.2) {
        for (int i = 0; i < 100; ++i) launch();
        (void)hipDeviceSynchronize();
        launches += 100;
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long st[2048][4];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
    std::vector<double> mhz, us;
    for (int w = 0; w < batch && w < 2048; ++w)
        if (st[w][3] > st[w][1]) {
            mhz.push_back(100.0 * (double)(st[w][2] - st[w][0]) / (double)(st[w][3] - st[w][1]));
            us.push_back((double)(st[w][3] - st[w][1]) * 0.01);
        }
    std::sort(mhz.begin(), mhz.end());
    std::sort(us.begin(), us.end());
    if (mhz.empty()) { printf("no stamps\n"); return 1; }
    printf("[%d output buffers in turn] k_supersaw_wide<4>, %d instances x %d voices x %lld frames: %.1f us per launch (events, 20 launches after %d), a workgroup's "
           "body %.1f us (median), in-kernel clock %.0f MHz (median; min %.0f, max %.0f)\n", ring, batch, nv, (long long)n, ms * 1e3 / 20,
           launches, us[us.size() / 2], mhz[mhz.size() / 2], mhz.front(), mhz.back());
    return 0;
}
