// pgx_convolve.hip -- ConvolvePE: streaming linear convolution y = x * h as a dense
// Toeplitz(h) x Hankel(x) product on the f32 MFMA units (convolve_pe.py:250-342).
//
// The reference evaluates y[i] = sum_k h[k] x[i-k] by float64 FFT overlap-save.  Here the same
// sum is a GEMM whose operands are generated from the two 1-D arrays, so HBM traffic is only
// x, h and y and the kernel is bound by the matrix pipe:
//
//   outputs i = 256*T + 16*q + p  (tile T, column q, row p);  GEMM index j = k + 15 - p
//       Y[p][q] = sum_j A[p][j] * B[j][q],   A[p][j] = h[j - 15 + p],   B[j][q] = e[256T + 16q + L-1 + 15 - j]
//   where e = [history (L-1 samples) | current block].  A depends on (p, j) only and B on (j, q)
//   only, which is exactly the v_mfma_f32_16x16x4_f32 operand shape (A[l&15][l>>4], B[l>>4][l&15]).
//
// Layout / staging
//   * prep kernel: e is written to the workspace in a 16-way POLYPHASE layout eP[c][r][Q]
//     (index = 16Q + r) with zero padding on both sides; h is written planar and zero padded.
//     With that layout a B fragment (16 consecutive q at one residue r) is 16 consecutive
//     floats, so LDS reads are conflict-free and LDS fills are straight row copies.
//   * main kernel: a 256-thread workgroup owns 16 tiles (4096 outputs) of one channel and a range
//     of K-chunks (KC taps each).  Per chunk it stages KC+32 taps and the matching e window
//     (16 rows x ~(KC+4096)/16) in LDS; every wave then runs 4 independent 16x16 accumulators
//     (4 tiles sharing one A fragment): per K-slab 1 A read + 4 B reads + 4 MFMAs.
//   * numerics: f32 MFMA is an exact-product fmaf chain; every 1024 taps the f32 accumulators are
//     folded into float64 registers, K-chunk partials are float64 in the workspace and the
//     reduce kernel sums them in a fixed order -> deterministic, error ~1e-7 of peak.
//   * history: the last L-1 inputs are copied back from the e image after the product.

#include "pgx_common.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kCvBlock = 256;
constexpr int kCvWaves = 4;
constexpr int kCvNT = 4;                        // tiles per wave
constexpr int kCvTiles = kCvWaves * kCvNT;      // 16 tiles per workgroup
constexpr int kCvOut = kCvTiles * 256;          // 4096 outputs per workgroup
constexpr int kCvFlush = 256;                   // K-slabs between float64 folds (1024 taps)

template <int KC>
struct CvGeom {
    static constexpr int kHs = KC + 32;                                  // staged taps (15 zero each side)
    static constexpr int kSlabs = (KC + 15 + 3) / 4;                     // K-slabs per chunk
    static constexpr int kSpan = kCvOut - 16 + 4 * kSlabs + 16;          // e indices touched
    static constexpr int kRows = kSpan / 16 + 2;                         // Q rows per residue
    static constexpr int kRowStride = ((kRows + 31) / 32) * 32 + 16;     // == 16 (mod 32)
};

struct CvPlan {
    int kc;             // taps per chunk
    int n_chunks;       // ceil(L / kc)
    int n_split;        // workgroups along K
    int chunks_per_split;
    int tile_groups;    // ceil(n / 4096)
    int64_t npad;       // tile_groups * 4096
    int64_t fp;         // front padding of the e image (multiple of 16)
    int64_t ql;         // Q rows of the polyphase image
    int64_t lp;         // padded planar h length
    size_t off_e, off_h, off_partial, total;
};

CvPlan cv_plan(int64_t n, int64_t L, int out_ch) {
    CvPlan p;
    p.kc = (L <= 1024) ? 512 : 4096;
    p.n_chunks = (int)pgx::ceil_div(L, p.kc);
    p.tile_groups = (int)pgx::ceil_div(n, kCvOut);
    p.npad = (int64_t)p.tile_groups * kCvOut;
    int64_t wg = (int64_t)p.tile_groups * out_ch;
    int want = (int)pgx::ceil_div(3072, wg);
    if (want < 1) want = 1;
    if (want > p.n_chunks) want = p.n_chunks;
    p.chunks_per_split = (int)pgx::ceil_div(p.n_chunks, want);
    p.n_split = (int)pgx::ceil_div(p.n_chunks, p.chunks_per_split);
    int64_t ktot = (int64_t)p.n_chunks * p.kc;
    p.fp = ((ktot + 64 + 15) / 16) * 16;
    int64_t span = p.fp + (L - 1) + p.npad + 64;
    p.ql = span / 16 + 2 + 1100;                    // + one LDS window of slack for tail reads
    p.lp = ktot + 64;
    size_t e_bytes = (size_t)out_ch * 16 * p.ql * sizeof(float);
    size_t h_bytes = (size_t)out_ch * p.lp * sizeof(float);
    size_t p_bytes = (size_t)p.n_split * out_ch * p.npad * sizeof(double);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    p.off_e = 0;
    p.off_h = up(e_bytes);
    p.off_partial = p.off_h + up(h_bytes);
    p.total = p.off_partial + up(p_bytes);
    return p;
}

// ---------------------------------------------------------------------------------------------
// prep: polyphase e image + planar h
__global__ void __launch_bounds__(256)
k_conv_prep(float *eP, float *hp, const float *x, const float *hist, const float *h, int64_t n,
            int src_ch, int64_t L, int fir_ch, int out_ch, int64_t fp, int64_t ql, int64_t lp) {
    const int64_t Lm1 = L - 1;
    const int64_t e_total = (int64_t)out_ch * 16 * ql;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < e_total; t += stride) {
        int64_t Q = t % ql;
        int64_t rc = t / ql;
        int r = (int)(rc % 16);
        int c = (int)(rc / 16);
        int64_t m = 16 * Q + r - fp;                 // index into e = [hist | x]
        float v = 0.0f;
        if (m >= 0 && m < Lm1) v = hist[m * out_ch + c];
        else if (m >= Lm1 && m < Lm1 + n) v = x[(m - Lm1) * src_ch + (src_ch == 1 ? 0 : c)];
        eP[t] = v;
    }
    const int64_t h_total = (int64_t)out_ch * lp;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < h_total; t += stride) {
        int64_t k = t % lp;
        int c = (int)(t / lp);
        hp[t] = (k < L) ? h[k * fir_ch + (fir_ch == 1 ? 0 : c)] : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
template <int KC>
__global__ void __launch_bounds__(kCvBlock)
k_conv_mfma(double *partial, const float *eP, const float *hp, int64_t L, int64_t fp, int64_t ql,
            int64_t lp, int64_t npad, int n_chunks, int chunks_per_split) {
    using G = CvGeom<KC>;
    __shared__ float hs[G::kHs];
    __shared__ float es[16 * G::kRowStride];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tg = blockIdx.x, ks = blockIdx.y, c = blockIdx.z;
    const int p16 = lane & 15, kk = lane >> 4;
    const int64_t Lm1 = L - 1;
    const float *eC = eP + (int64_t)c * 16 * ql;
    const float *hC = hp + (int64_t)c * lp;

    double dacc[kCvNT][4];
#pragma unroll
    for (int t = 0; t < kCvNT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dacc[t][r] = 0.0;

    const int kc_begin = ks * chunks_per_split;
    int kc_end = kc_begin + chunks_per_split;
    if (kc_end > n_chunks) kc_end = n_chunks;

    for (int kc = kc_begin; kc < kc_end; ++kc) {
        const int64_t k0 = (int64_t)kc * KC;
        // ---- stage taps: hs[idx] = h[k0 + idx - 15] for idx in [15, KC+15), else 0 ----
        for (int idx = tid; idx < G::kHs; idx += kCvBlock) {
            float v = 0.0f;
            if (idx >= 15 && idx < KC + 15) v = hC[k0 + idx - 15];      // hp is zero padded past L
            hs[idx] = v;
        }
        // ---- stage the e window, polyphase rows ----
        // lowest e' index touched by this workgroup in this chunk (tile 0, q 0, last slab)
        const int64_t lo = (int64_t)256 * kCvTiles * tg + Lm1 + 15 - k0 - (4 * G::kSlabs - 1) + fp;
        const int64_t qbase = lo >> 4;                                   // lo >= 0 by construction of fp
        for (int t = tid; t < 16 * G::kRows; t += kCvBlock) {
            int r = t / G::kRows, ql_i = t - r * G::kRows;
            int64_t Q = qbase + ql_i;
            es[r * G::kRowStride + ql_i] = (Q < ql) ? eC[(int64_t)r * ql + Q] : 0.0f;
        }
        __syncthreads();

        // ---- MFMA over the chunk ----
        // K-slabs are walked in groups of 4 (16 taps): within a group the e' residue pattern of a lane
        // is loop invariant, so the per-slab address arithmetic collapses to one subtraction, and the
        // operands of group g+1 are fetched from LDS while the 16 MFMAs of group g issue.
        floatx4 acc[kCvNT];
#pragma unroll
        for (int t = 0; t < kCvNT; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
        // local e' index (relative to 16*qbase) of tile (4*wave), q = 0, at j-local 0, kk of this lane
        const int vb = (int)((int64_t)256 * (kCvTiles * tg + kCvNT * wave) + Lm1 + 15 - k0 + fp - 16 * qbase) - kk;
        int off[4];                       // es index of slab j of group 0 (per lane, loop invariant)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int vj = vb - 4 * j;
            off[j] = (vj & 15) * G::kRowStride + (vj >> 4) + p16;
        }
        const float *ap = hs + kk + p16;  // A fragment of slab 0; +16 floats per group
        constexpr int kGroups = G::kSlabs / 4;
        static_assert(G::kSlabs % 4 == 0, "K-slabs per chunk must be a multiple of 4");

        float a_cur[4], b_cur[4][kCvNT], a_nxt[4], b_nxt[4][kCvNT];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a_cur[j] = ap[4 * j];
#pragma unroll
            for (int t = 0; t < kCvNT; ++t) b_cur[j][t] = es[off[j] + 16 * t];
        }
        for (int g0 = 0; g0 < kGroups; g0 += kCvFlush / 4) {
            const int g1 = (g0 + kCvFlush / 4 < kGroups) ? g0 + kCvFlush / 4 : kGroups;
            for (int g = g0; g < g1; ++g) {
                const bool more = (g + 1 < kGroups);
                if (more) {
                    ap += 16;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        a_nxt[j] = ap[4 * j];
                        const float *bp = es + (off[j] - (g + 1));          // v drops by 16 per group: Q - 1
#pragma unroll
                        for (int t = 0; t < kCvNT; ++t) b_nxt[j][t] = bp[16 * t];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < kCvNT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j], b_cur[j][t], acc[t], 0, 0, 0);
                if (more) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        a_cur[j] = a_nxt[j];
#pragma unroll
                        for (int t = 0; t < kCvNT; ++t) b_cur[j][t] = b_nxt[j][t];
                    }
                }
            }
            // fold the f32 accumulators into float64 every kCvFlush slabs (1024 taps)
#pragma unroll
            for (int t = 0; t < kCvNT; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dacc[t][r] += (double)acc[t][r];
                acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();       // LDS is overwritten by the next chunk
    }

    // ---- write float64 partials: C/D layout col(q) = lane&15, row(p) = 4*(lane>>4) + reg ----
    double *out = partial + ((int64_t)ks * gridDim.z + c) * npad;
#pragma unroll
    for (int t = 0; t < kCvNT; ++t) {
        const int64_t i0 = (int64_t)256 * (kCvTiles * tg + kCvNT * wave + t) + 16 * p16 + 4 * kk;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[i0 + r] = dacc[t][r];
    }
}

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_conv_reduce(float *out, const double *partial, int64_t n, int out_ch, int64_t npad, int n_split) {
    const int64_t total = n * out_ch;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        int64_t i = e / out_ch;
        int c = (int)(e - i * out_ch);
        double acc = 0.0;
        for (int s = 0; s < n_split; ++s) acc += partial[((int64_t)s * out_ch + c) * npad + i];
        out[e] = (float)acc;
    }
}

// hist[j][c] = e[n + j] for j in [0, L-1): the last L-1 samples of [hist | x] (convolve_pe.py:323-337)
__global__ void __launch_bounds__(256)
k_conv_hist(float *hist, const float *eP, int64_t n, int64_t L, int out_ch, int64_t fp, int64_t ql) {
    const int64_t total = (L - 1) * out_ch;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        int64_t j = e / out_ch;
        int c = (int)(e - j * out_ch);
        int64_t m = n + j + fp;
        hist[e] = eP[((int64_t)c * 16 + (m & 15)) * ql + (m >> 4)];
    }
}

}  // namespace

extern "C" {

size_t pgx_convolve_workspace_bytes(int64_t n, int64_t fir_len, int out_channels) {
    if (n <= 0 || fir_len <= 0 || out_channels <= 0) return 0;
    return cv_plan(n, fir_len, out_channels).total;
}

int pgx_convolve(float *out, const float *x, int64_t n, int src_channels, const float *h, int64_t fir_len,
                 int fir_channels, int out_channels, float *hist, void *workspace) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && x && h && workspace, "pgx_convolve: null pointer");
    PGX_CHECK_ARG(fir_len >= 1 && src_channels >= 1 && fir_channels >= 1 && out_channels >= 1,
                  "pgx_convolve: bad shape");
    PGX_CHECK_ARG(fir_len == 1 || hist != nullptr, "pgx_convolve: history buffer required");
    PGX_CHECK_ARG((src_channels == 1 || src_channels == out_channels) &&
                      (fir_channels == 1 || fir_channels == out_channels),
                  "pgx_convolve: channel counts must be 1 or equal to out_channels");
    PGX_CHECK_ARG(out_channels <= 65535, "pgx_convolve: too many channels");
    CvPlan p = cv_plan(n, fir_len, out_channels);
    char *ws = (char *)workspace;
    float *eP = (float *)(ws + p.off_e);
    float *hp = (float *)(ws + p.off_h);
    double *partial = (double *)(ws + p.off_partial);
    hipStream_t st = pgx::stream();

    int64_t prep_items = (int64_t)out_channels * 16 * p.ql;
    hipLaunchKernelGGL(k_conv_prep, dim3(pgx::grid_for(prep_items, 256)), dim3(256), 0, st, eP, hp, x,
                       (const float *)hist, h, n, src_channels, fir_len, fir_channels, out_channels, p.fp, p.ql,
                       p.lp);
    PGX_LAUNCH_CHECK("k_conv_prep");
    dim3 grid(p.tile_groups, p.n_split, out_channels);
    if (p.kc == 512) {
        hipLaunchKernelGGL(k_conv_mfma<512>, grid, dim3(kCvBlock), 0, st, partial, (const float *)eP,
                           (const float *)hp, fir_len, p.fp, p.ql, p.lp, p.npad, p.n_chunks, p.chunks_per_split);
    } else {
        hipLaunchKernelGGL(k_conv_mfma<4096>, grid, dim3(kCvBlock), 0, st, partial, (const float *)eP,
                           (const float *)hp, fir_len, p.fp, p.ql, p.lp, p.npad, p.n_chunks, p.chunks_per_split);
    }
    PGX_LAUNCH_CHECK("k_conv_mfma");
    hipLaunchKernelGGL(k_conv_reduce, dim3(pgx::grid_for(n * out_channels, 256)), dim3(256), 0, st, out,
                       (const double *)partial, n, out_channels, p.npad, p.n_split);
    PGX_LAUNCH_CHECK("k_conv_reduce");
    if (fir_len > 1) {
        hipLaunchKernelGGL(k_conv_hist, dim3(pgx::grid_for((fir_len - 1) * out_channels, 256)), dim3(256), 0, st,
                           hist, (const float *)eP, n, fir_len, out_channels, p.fp, p.ql);
        PGX_LAUNCH_CHECK("k_conv_hist");
    }
    return PGX_OK;
}

}  // extern "C"
