// What k_adsr_walk_par decides for C5's envelopes (GPU box): starts per envelope, verification rounds, fallbacks, and the
// kernel's time by HIP events -- the library's source with a debug hook, driven through its own entry point.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I include -I pygmu2_amd/csrc tools/microbench/adsr_par_debug.hip -o /tmp/adsr_par_debug
#include <hip/hip_runtime.h>
__device__ int g_dbg[1024][8];
#define PGX_ADSR_DEBUG(slot, value)                                             \
    do {                                                                        \
        if (threadIdx.x == 0) g_dbg[blockIdx.x][(slot)] = (int)(value);         \
    } while (0)
#define PGX_ADSR_CLOCK() wall_clock64()
#define PGX_ADSR_DEBUG_MAX(slot, value)                                         \
    do {                                                                        \
        if ((threadIdx.x & 63) == 0) atomicMax(&g_dbg[blockIdx.x][(slot)], (int)(value));   \
    } while (0)
#include "../../pygmu2_amd/csrc/pgx_adsr.hip"
#include <cmath>
#include <cstdio>
#include <vector>
static hipStream_t g_stream;
namespace pgx {
static thread_local std::string g_err;
void set_error(const std::string &m) { g_err = m; }
int fail(int code, const std::string &m) { g_err = m; fprintf(stderr, "fail: %s\n", m.c_str()); return code; }
hipStream_t stream() { return g_stream; }
hipStream_t main_stream() { return g_stream; }
bool initialised() { return true; }
int device_index() { return 0; }
}  // namespace pgx
extern "C" int pgx_memset(void *p, int v, size_t n) { return hipMemsetAsync(p, v, n, g_stream) == hipSuccess ? 0 : -1; }
extern "C" int pgx_stream_fork(void) { return 0; }
extern "C" int pgx_stream_select(int) { return 0; }

int main(int argc, char **argv) {
    const int batch = argc > 1 ? atoi(argv[1]) : 64, stride8 = 512 / batch;
    const int64_t n = 48000;
    hipStreamCreate(&g_stream);
    std::vector<pgx_gate_params> hg(batch);
    std::vector<pgx_adsr_params> hp(batch);
    for (int v = 0; v < batch; ++v) {
        const double f = 2.0 + 0.01 * (v * stride8);
        hg[v] = pgx_gate_params{f / 48000.0, 0.0, 0.5};
        hp[v] = pgx_adsr_params{1.0 / (0.01 * 48000.0), (0.7 - 1.0) / (0.1 * 48000.0), -0.7 / (0.2 * 48000.0), 0.7, 0};
    }
    pgx_gate_params *dg; pgx_adsr_params *dp; double *st; float *out; void *ws;
    hipMalloc(&dg, batch * sizeof(hg[0])); hipMalloc(&dp, batch * sizeof(hp[0]));
    hipMalloc(&st, batch * 3 * 8); hipMemset(st, 0, batch * 3 * 8);
    hipMalloc(&out, batch * n * 4);
    const size_t wsb = pgx_adsr_workspace_bytes(batch, n);
    hipMalloc(&ws, wsb);
    hipMemcpy(dg, hg.data(), batch * sizeof(hg[0]), hipMemcpyHostToDevice);
    hipMemcpy(dp, hp.data(), batch * sizeof(hp[0]), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blk = 0; blk < 6; ++blk) {
        hipEventRecord(e0, g_stream);
        pgx_adsr_gated_periodic(out, n, batch, blk * n, n, dg, dp, st, ws, 0);
        hipEventRecord(e1, g_stream);
        hipStreamSynchronize(g_stream);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        static int dbg[1024][8];
        hipMemcpyFromSymbol(dbg, HIP_SYMBOL(g_dbg), sizeof(dbg));
        int hist_ns[16] = {0}, gave = 0, rounds_max = 0, spec = 0, t5 = 0, t6 = 0, t7 = 0, v7 = 0;
        for (int v = 0; v < batch; ++v) {
            hist_ns[dbg[v][0] & 15]++; gave += dbg[v][1]; spec += dbg[v][3];
            if (dbg[v][4] > rounds_max) rounds_max = dbg[v][4];
            if (dbg[v][5] > t5) t5 = dbg[v][5];
            if (dbg[v][6] > t6) t6 = dbg[v][6];
            if (dbg[v][7] > t7) { t7 = dbg[v][7]; v7 = v; }
        }
        printf("  slowest: list %.2f us, verify %.2f us, emit %.2f us (envelope %d)\n", t5 * 0.01, t6 * 0.01, t7 * 0.01, v7);
        if (blk == 3) {
            printf("  rounds per envelope:");
            for (int v = 0; v < batch; ++v) printf(" %d", dbg[v][4]);
            printf("\n  starts per envelope:");
            for (int v = 0; v < batch; ++v) printf(" %d", dbg[v][0]);
            printf("\n  verify us per envelope:");
            for (int v = 0; v < batch; ++v) printf(" %.0f", dbg[v][6] * 0.01);
            printf("\n  emit us per envelope:");
            for (int v = 0; v < batch; ++v) printf(" %.0f", dbg[v][7] * 0.01);
            printf("\n");
        }
        static int zero[1024][8];
        hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), zero, sizeof(zero));
        printf("block %d: edges + walk %.1f us; starts per envelope:", blk, ms * 1e3);
        for (int k = 1; k <= 8; ++k) printf(" %d:%d", k, hist_ns[k]);
        printf("; with speculative starts %d, fallbacks %d, most rounds %d; edges of v0 / v%d: %d / %d\n", spec, gave, rounds_max,
               batch - 1, dbg[0][2], dbg[batch - 1][2]);
    }
    return 0;
}
