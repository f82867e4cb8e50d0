// Where do the cycles of the row pass of the FFT convolution go?  A copy of k_fft_rows<true, 1024> with
// s_memtime stamps between its phases, launched in the C3 shape (128 x 2 workgroups).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I include -I pygmu2_amd/csrc tools/microbench/fft_phases.hip -o /tmp/fft_phases
#include "../../pygmu2_amd/csrc/pgx_fftconv.hip"

namespace pgx {
static thread_local std::string g_err;
void set_error(const std::string &m) { g_err = m; }
int fail(int code, const std::string &m) { g_err = m; return code; }
hipStream_t stream() { return nullptr; }
hipStream_t main_stream() { return nullptr; }
bool initialised() { return true; }
int device_index() { return 0; }
}  // namespace pgx

namespace {
constexpr int kStamps = 10;
template <int TILE>
__global__ void __launch_bounds__(kFBlock)
k_rows_timed(cplx *work, ConvGeom g, Tables tb, const cplx *H, long long *stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PT = TILE / kFBlock;
    const int N2 = (int)g.N2;
    cplx *buf = reinterpret_cast<cplx *>(smem);
    cplx *alt = buf + TILE;
    cplx *tw = alt + TILE;
    const int tid = threadIdx.x;
    const int64_t pair = blockIdx.y;
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    cplx *wk = work + pair * g.N + tile0;
    long long t[kStamps];
    int ns = 0;
    t[ns++] = wall_clock64();
    cplx v[PT], hv[PT], bigtw[PT];
#pragma unroll
    for (int u = 0; u < PT; ++u) v[u] = wk[tid + u * kFBlock];
    const cplx *Hc = H + tile0;
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        hv[u] = Hc[tid + u * kFBlock];
        bigtw[u] = tb.big[tile0 + tid + u * kFBlock];
    }
    fill_stage_twiddles(tw, tb.t2, g.l2);
    t[ns++] = wall_clock64();
#pragma unroll
    for (int u = 0; u < PT; ++u) buf[tid + u * kFBlock] = v[u];
    __syncthreads();
    t[ns++] = wall_clock64();
    cplx *res = lds_fft<TILE>(buf, alt, tw, g.l2, N2);
    t[ns++] = wall_clock64();
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int e = tid + u * kFBlock;
        res[e] = cconj(cmul(res[e], hv[u]));
    }
    __syncthreads();
    t[ns++] = wall_clock64();
    const cplx *fin = lds_fft<TILE>(res, res == buf ? alt : buf, tw, g.l2, N2);
    t[ns++] = wall_clock64();
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int e = tid + u * kFBlock;
        wk[e] = cmul(cconj(fin[e]), cconj(bigtw[u]));
    }
    __threadfence();
    t[ns++] = wall_clock64();
    if (tid == 0)
        for (int i = 0; i < ns; ++i) stamps[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * kStamps + i] = t[i];
}
}  // namespace

int main() {
    ConvGeom g{};
    fft_geometry(131072, 65536, g);
    g.n = 96000; g.nblocks = 2; g.npairs = 1; g.src_ch = 2; g.out_ch = 2;
    const int pairs = 2;
    cplx *work, *spec;
    long long *stamps;
    hipMalloc(&work, sizeof(cplx) * g.N * pairs);
    hipMemset(work, 0, sizeof(cplx) * g.N * pairs);
    const size_t spec_elems = g.N + g.N + g.N1 + g.N2;
    hipMalloc(&spec, sizeof(cplx) * spec_elems);
    hipMemset(spec, 0, sizeof(cplx) * spec_elems);
    hipMalloc(&stamps, sizeof(long long) * kStamps * 256);
    const Tables tb = tables_of(spec, g, 1);
    hipLaunchKernelGGL(k_fft_tables, dim3(512), dim3(kFBlock), 0, 0, const_cast<cplx *>(tb.big), const_cast<cplx *>(tb.t1),
                       const_cast<cplx *>(tb.t2), g);
    hipDeviceSynchronize();
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);      // kHz
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_rows_timed<1024>), dim3(128, pairs), dim3(kFBlock), rows_smem(g, 1024), 0, work, g, tb, spec, stamps);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    static long long h[kStamps * 256];
    hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[] = {"global loads + twiddle fill", "to LDS + barrier", "forward FFT (5 stages)", "x H + barrier",
                           "inverse FFT (5 stages)", "twiddle + store"};
    printf("kernel (events): %.2f us; wall clock %d kHz\n", ms * 1e3, rate);
    for (int wg : {0, 1, 77, 200}) {
        printf("workgroup %3d:", wg);
        for (int i = 0; i + 1 < 7; ++i) printf("  %s %.2f us;", names[i], (h[wg * kStamps + i + 1] - h[wg * kStamps + i]) / (rate * 1e-3));
        printf("\n");
    }
    long long lo = h[0], hi = h[6];
    for (int w = 0; w < 256; ++w) { if (h[w * kStamps] < lo) lo = h[w * kStamps]; if (h[w * kStamps + 6] > hi) hi = h[w * kStamps + 6]; }
    printf("first start -> last end over all workgroups: %.2f us\n", (hi - lo) / (rate * 1e-3));
    return 0;
}
