// Where do the three launches of pgx_convolve_fft spend a call?  The library's own kernels (source included) with
// wall_clock64() stamps (100 MHz) between their phases: C3's 96 000-frame stereo call by default.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I include -I pygmu2_amd/csrc tools/microbench/fft_phases.hip -o tools/microbench/fft_phases
#include <hip/hip_runtime.h>
__device__ long long g_stamps[5][4096][8];
#define PGX_FFT_STAMP(k, i)                                                                               \
    do {                                                                                                  \
        if (threadIdx.x == 0) {                                                                           \
            const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                     \
            if (wg_ < 4096) g_stamps[(k)][wg_][(i)] = wall_clock64();                                     \
        }                                                                                                 \
    } while (0)
#include "../../pygmu2_amd/csrc/pgx_fftconv.hip"

#include <cmath>
#include <cstdio>
#include <vector>

namespace pgx {
static thread_local std::string g_err;
void set_error(const std::string &m) { g_err = m; }
int fail(int code, const std::string &m) { g_err = m; fprintf(stderr, "pgx error: %s\n", m.c_str()); return code; }
hipStream_t stream() { return nullptr; }
hipStream_t main_stream() { return nullptr; }
bool initialised() { return true; }
int device_index() { return 0; }
}  // namespace pgx

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 96000, L = 65536;
    const int64_t nfft = argc > 2 ? atoll(argv[2]) : 131072;
    const int reps = 6;
    std::vector<float> hx(n * 2), hh(L);
    for (int64_t i = 0; i < n * 2; ++i) hx[i] = 0.1f * (float)std::sin(0.37 * i);
    for (int64_t i = 0; i < L; ++i) hh[i] = (float)(std::cos(1.3 * i) * std::exp(-i / 8000.0));
    float *x, *h, *out, *hist;
    void *spec, *ws;
    const size_t sb = pgx_convolve_fft_spectrum_bytes(nfft, 1), wb = pgx_convolve_fft_workspace_bytes(n, L, 2, nfft);
    hipMalloc(&x, hx.size() * 4);
    hipMalloc(&h, hh.size() * 4);
    hipMalloc(&out, hx.size() * 4);
    hipMalloc(&hist, (L - 1) * 2 * 4);
    hipMalloc(&spec, sb);
    hipMalloc(&ws, wb);
    hipMemset(hist, 0, (L - 1) * 2 * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(h, hh.data(), hh.size() * 4, hipMemcpyHostToDevice);
    if (pgx_convolve_fft_prepare(spec, h, L, 1, nfft)) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < reps; ++rep) {
        if (rep == 1) hipEventRecord(e0, nullptr);
        if (pgx_convolve_fft(out, x, n, 2, spec, L, 1, 2, nfft, hist, ws, 0)) return 1;
    }
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    static long long st[5][4096][8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
    const int64_t V = nfft - (L - 1), nblocks = (n + V - 1) / V, pairs = (nblocks * 2 + 1) / 2;
    const int tile = nfft >= (1 << 18) ? 2048 : 1024;
    const int wgs = (int)std::min<int64_t>(4096, nfft / tile * pairs);
    printf("n %lld nfft %lld: %lld transforms, %d workgroups per launch, %.2f us per call by events\n", (long long)n,
           (long long)nfft, (long long)pairs, wgs, ms * 1000.0 / (reps - 1));
    long long t0 = st[0][0][0];
    for (int w = 0; w < wgs; ++w) t0 = std::min(t0, st[0][w][0]);
    const struct { int k, last; const char *name; const char *phases[5]; } ks[3] = {
        {0, 3, "k_fft_cols<0>", {"loads+table", "fft", "twiddle+store", "", ""}},
        {4, 5, "k_fft_rows   ", {"loads+table", "fft", "xH+fft", "twiddle+store", "history"}},
        {2, 3, "k_fft_cols<2>", {"loads+table", "fft", "store", "", ""}}};
    for (const auto &k : ks) {
        double s_min = 1e30, s_avg = 0, s_max = 0, e_max = 0, ph[5] = {0};
        for (int w = 0; w < wgs; ++w) {
            const long long *r = st[k.k][w];
            const double s = (r[0] - t0) * 0.01, e = (r[k.last] - t0) * 0.01;
            s_min = std::min(s_min, s);
            s_max = std::max(s_max, s);
            s_avg += s / wgs;
            e_max = std::max(e_max, e);
            for (int p = 0; p < k.last; ++p) ph[p] += (r[p + 1] - r[p]) * 0.01 / wgs;
        }
        printf(" %s: workgroups start %.2f .. %.2f (mean %.2f) us, last one ends %.2f us;", k.name, s_min, s_max, s_avg, e_max);
        for (int p = 0; p < k.last; ++p) printf("  %s %.2f", k.phases[p], ph[p]);
        printf("\n");
    }
    return 0;
}
