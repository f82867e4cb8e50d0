// Where does a workgroup of the time-segmented SuperSaw bank spend its life?  k_supersaw_bank<8> itself (the library's
// source, included) with wall_clock64() stamps (100 MHz) between its phases, launched in the shape of a rank's share
// at G = 8: 64 instances x 4 segments of a 48 000-frame block.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I include -I pygmu2_amd/csrc tools/microbench/ss_phases.hip -o /tmp/ss_phases
#include <hip/hip_runtime.h>
__device__ long long g_stamps[1024][16];
#define PGX_SS_STAMP(i)                                                                         \
    do {                                                                                        \
        if (threadIdx.x == 0) g_stamps[blockIdx.y * gridDim.x + blockIdx.x][(i)] = wall_clock64(); \
    } while (0)
#include "../../pygmu2_amd/csrc/pgx_scan.hip"

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
    const int batch = argc > 1 ? atoi(argv[1]) : 64, nseg = argc > 2 ? atoi(argv[2]) : 4, nv = 7;
    const int64_t n = argc > 3 ? atoll(argv[3]) : 48000;
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
    double *ds, *ds2, *da, *dt;
    float *out;
    hipMalloc(&dp, hp.size() * sizeof(hp[0]));
    hipMalloc(&ds, hs.size() * 8);
    hipMalloc(&ds2, hs.size() * 8);
    hipMalloc(&da, ha.size() * 8);
    hipMalloc(&dt, (size_t)batch * nv * kSsTabDoubles * 8);
    hipMalloc(&out, (size_t)batch * n * 4);
    hipMemcpy(dp, hp.data(), hp.size() * sizeof(hp[0]), hipMemcpyHostToDevice);
    hipMemcpy(ds, hs.data(), hs.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(da, ha.data(), ha.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_supersaw_tables, dim3(batch), dim3(256), 0, 0, dt, nv, 48000.0, dp);
    const int64_t tiles = (n + 4095) / 4096;
    const int seg_tiles = (int)((tiles + nseg - 1) / nseg);
    const bool wide = argc > 4 && atoi(argv[4]) != 0;        // 4th argument 1: k_supersaw_wide (16 frames per thread)
    int seg_tiles_w = 0;
    if (wide) {
        double *dtw;
        hipMalloc(&dtw, (size_t)batch * nv * kSswTabDoubles * 8);
        hipLaunchKernelGGL(k_supersaw_wide_tables, dim3(batch), dim3(64), 0, 0, dtw, nv, 48000.0, dp);
        const int64_t tiles_w = (n + 4095) / 4096;
        seg_tiles_w = (int)((tiles_w + nseg - 1) / nseg);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_supersaw_wide<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((size_t)nv * (256 * 32 + kSswTabDoubles * 8)));
        for (int rep = 0; rep < 3; ++rep)
            hipLaunchKernelGGL(k_supersaw_wide<4>, dim3(batch, nseg), dim3(256), (size_t)nv * (256 * 32 + kSswTabDoubles * 8), 0, out, n, nv, n, 1,
                               (const double *)ds, ds2, (const double *)da, seg_tiles_w, (const double *)dtw, 1);
    } else
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL(k_supersaw_bank<8>, dim3(batch, nseg), dim3(512), 0, 0, out, n, nv, n, 1, 48000.0, dp,
                           (const double *)ds, ds2, (const double *)da, seg_tiles, (const double *)dt);
    hipDeviceSynchronize();
    static long long st[1024][16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
    long long t0 = st[0][0], t1 = 0;
    for (int w = 0; w < batch * nseg; ++w) {
        if (st[w][0] < t0) t0 = st[w][0];
        if (st[w][15] > t1) t1 = st[w][15];
    }
    printf("batch %d segments %d n %lld: first start -> last end %.2f us\n", batch, nseg, (long long)n, (t1 - t0) * 0.01);
    for (int sgm = 0; sgm < nseg; ++sgm) {
        double acc[16] = {0}, start = 0, end = 0;
        for (int i = 0; i < batch; ++i) {
            const long long *r = st[sgm * batch + i];
            start += (r[0] - t0) * 0.01;
            end += (r[15] - t0) * 0.01;
            acc[1] += (r[1] - r[0]) * 0.01;
            acc[2] += (r[2] - r[1]) * 0.01;
            for (int t = 0; t < seg_tiles; ++t) {
                const long long nxt = (t + 1 < seg_tiles && r[3 + t + 1] > r[3 + t]) ? r[3 + t + 1] : r[15];
                acc[3 + t] += (nxt - r[3 + t]) * 0.01;
            }
        }
        printf(" segment %d: starts at %.2f us, tables %.2f, carries %.2f, tiles", sgm, start / batch, acc[1] / batch,
               acc[2] / batch);
        for (int t = 0; t < seg_tiles; ++t) printf(" %.2f", acc[3 + t] / batch);
        printf(", ends at %.2f us\n", end / batch);
    }
    return 0;
}
