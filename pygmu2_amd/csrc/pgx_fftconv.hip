// pgx_fftconv.hip -- ConvolvePE for long filters: float64 FFT overlap-save (convolve_pe.py:250-342).
//
// The reference itself evaluates the convolution by FFT overlap-save in float64 (numpy pocketfft);
// for tens of thousands of taps that is 3-4 orders of magnitude fewer operations than the direct
// form the MFMA path (pgx_convolve.hip) evaluates, so long filters come here.
//
// Structure (all float64, one launch each, no transposes):
//   N = N1*N2 point FFT by the four-step decomposition  i = i1*N2 + i2,  k = k1 + N1*k2
//   (1) k_fft_cols<fwd>   packs two real input blocks (overlap-save blocks 2p and 2p+1 of one output
//                         channel) into re/im of one complex sequence, length-N1 FFTs down the columns,
//                         twiddle W_N^(i2*k1)                                   -> work[k1][i2]
//   (2) k_fft_rows        length-N2 FFT along each row: S[k1][k2] = X[k1 + N1*k2] (a "scrambled" but
//                         fixed order), times the filter spectrum H kept in the same order, inverse
//                         row FFT, conjugate twiddle                            -> work[k1][i2]
//   (3) k_fft_cols<inv>   inverse column FFTs, scale 1/N; real part = block 2p, imaginary part = block
//                         2p+1 (the filter is real, so the two packed signals never mix); the valid
//                         overlap-save samples go straight to the float32 output.
// The spectrum is consumed in the order the forward transform produces it, so neither direction needs
// a transpose or a bit reversal.  Every workgroup transforms TILE = 1024 (2048 for N = 2^18) complex
// points held in LDS with a Stockham radix-4 (+ one radix-2) autosort FFT; twiddles come from a
// per-workgroup LDS table.  N2 = TILE (one row per workgroup), N1 = N / N2, so a column workgroup
// covers TILE / N1 >= 8 adjacent columns (128-byte rows) and a 2^17-point transform is 128 workgroups.
// HBM traffic per complex point: 16 B read + 16 B write per pass, three passes.

#include "pgx_common.h"

namespace {

constexpr int kFBlock = 256;
constexpr double kTwoPi = 6.283185307179586;

struct cplx {
    double x, y;
};
__device__ __forceinline__ cplx cmul(const cplx &a, const cplx &b) {
    return cplx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
__device__ __forceinline__ cplx cadd(const cplx &a, const cplx &b) { return cplx{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cplx csub(const cplx &a, const cplx &b) { return cplx{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cplx cconj(const cplx &a) { return cplx{a.x, -a.y}; }
__device__ __forceinline__ cplx mul_neg_i(const cplx &a) { return cplx{a.y, -a.x}; }      // a * (-i)

// W_n^m = exp(-2*pi*i*m/n), n a power of two (inv_n = 1/n is exact)
__device__ __forceinline__ cplx twiddle(int64_t m, double inv_n) {
    double sn, cs;
    pgx::pgx_sincos(kTwoPi * ((double)m * inv_n), sn, cs);
    return cplx{cs, -sn};
}

// Forward DFT of TILE/M sequences of length M = 2^lm stored at buf[s*stride + i], in place, natural
// order in and out.  tw[p] = W_M^p.  All 256 threads take part; ends with a barrier.
template <int TILE>
__device__ void lds_fft(cplx *buf, const cplx *tw, int lm, int stride) {
    const int tid = threadIdx.x;
    constexpr int U4 = TILE / 4 / kFBlock;                     // radix-4 butterflies per thread
    constexpr int U2 = TILE / 2 / kFBlock;                     // radix-2 butterflies per thread
    for (int lns = 0; lns < lm;) {
        const int Ns = 1 << lns;
        if (lm - lns >= 2) {
            const int lq = lm - 2, q = 1 << lq;                    // butterflies per sequence
            cplx r[U4][4];
#pragma unroll
            for (int u = 0; u < U4; ++u) {
                const int w = tid + u * kFBlock;
                const int s = w >> lq, j = w & (q - 1);
                const int k = j & (Ns - 1);
                const cplx *b = buf + s * stride;
                const int step = k << (lq - lns);                  // W_{4Ns}^(k*t) = W_M^(step*t)
                const cplx a = b[j];
                const cplx c1 = cmul(b[j + q], tw[step]);
                const cplx c2 = cmul(b[j + 2 * q], tw[2 * step]);
                const cplx c3 = cmul(b[j + 3 * q], tw[3 * step]);
                const cplx s0 = cadd(a, c2), s1 = csub(a, c2), s2 = cadd(c1, c3), s3 = mul_neg_i(csub(c1, c3));
                r[u][0] = cadd(s0, s2);
                r[u][1] = cadd(s1, s3);
                r[u][2] = csub(s0, s2);
                r[u][3] = csub(s1, s3);
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < U4; ++u) {
                const int w = tid + u * kFBlock;
                const int s = w >> lq, j = w & (q - 1);
                const int k = j & (Ns - 1);
                cplx *b = buf + s * stride + (j - k) * 4 + k;
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t * Ns] = r[u][t];
            }
            __syncthreads();
            lns += 2;
        } else {
            const int lq = lm - 1, q = 1 << lq;
            cplx r[U2][2];
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int w = tid + u * kFBlock;
                const int s = w >> lq, j = w & (q - 1);
                const int k = j & (Ns - 1);
                const cplx *b = buf + s * stride;
                const cplx a = b[j];
                const cplx c1 = cmul(b[j + q], tw[k << (lq - lns)]);
                r[u][0] = cadd(a, c1);
                r[u][1] = csub(a, c1);
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int w = tid + u * kFBlock;
                const int s = w >> lq, j = w & (q - 1);
                const int k = j & (Ns - 1);
                cplx *b = buf + s * stride + (j - k) * 2 + k;
                b[0] = r[u][0];
                b[Ns] = r[u][1];
            }
            __syncthreads();
            lns += 1;
        }
    }
}

__device__ __forceinline__ void fill_twiddles(cplx *tw, int M) {
    const double inv = 1.0 / (double)M;
    for (int p = threadIdx.x; p < M; p += kFBlock) tw[p] = twiddle(p, inv);
}

// Where the real sequences come from / go to.
struct ConvGeom {
    int64_t n;            // frames of the current block
    int64_t L;            // taps
    int64_t N, N1, N2;    // FFT geometry (powers of two)
    int l1, l2;           // log2 N1, log2 N2
    int64_t V;            // hop = N - (L - 1) valid outputs per overlap-save block
    int64_t nblocks;      // overlap-save blocks per channel
    int64_t npairs;       // ceil(nblocks / 2)
    int src_ch, out_ch;
};

// sample `pos` of overlap-save block `b` of output channel `ch`: (history | x) at b*V + pos
__device__ __forceinline__ double conv_input(const ConvGeom &g, const float *x, const float *hist, int64_t b, int ch,
                                             int64_t pos) {
    if (b >= g.nblocks) return 0.0;
    const int64_t e = b * g.V + pos;                               // index into (history | x)
    if (e < g.L - 1) return (double)hist[e * g.out_ch + ch];
    const int64_t i = e - (g.L - 1);
    if (i >= g.n) return 0.0;
    return (double)x[i * g.src_ch + (g.src_ch == 1 ? 0 : ch)];
}

// MODE 0: forward, input = packed signal blocks; MODE 1: forward, input = filter taps (spectrum
// preparation); MODE 2: inverse, output = float32 samples.
template <int MODE, int TILE>
__global__ void __launch_bounds__(kFBlock)
k_fft_cols(cplx *work, ConvGeom g, const float *x, const float *hist, const float *h, int fir_ch, float *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int N1 = (int)g.N1;
    const int CW = TILE >> g.l1, lcw = __builtin_ctz(CW);          // columns per workgroup
    const int stride = N1 + 1;                                     // +1: spread the columns over the banks
    cplx *buf = reinterpret_cast<cplx *>(smem);
    cplx *tw = buf + CW * stride;
    const int tid = threadIdx.x;
    const int64_t pair = blockIdx.y;
    const int64_t col0 = (int64_t)blockIdx.x * CW;
    cplx *wk = work + pair * g.N;
    const int ch = (MODE == 1) ? (int)pair : (int)(pair / g.npairs);
    const int64_t p = (MODE == 1) ? 0 : pair - (int64_t)ch * g.npairs;
    const double inv_n = 1.0 / (double)g.N;

    fill_twiddles(tw, N1);
    for (int e = tid; e < TILE; e += kFBlock) {
        const int c = e & (CW - 1), i1 = e >> lcw;
        const int64_t pos = ((int64_t)i1 << g.l2) + col0 + c;
        cplx v;
        if (MODE == 0) {
            v = cplx{conv_input(g, x, hist, 2 * p, ch, pos), conv_input(g, x, hist, 2 * p + 1, ch, pos)};
        } else if (MODE == 1) {
            v = cplx{pos < g.L ? (double)h[pos * fir_ch + ch] : 0.0, 0.0};
        } else {
            v = cconj(wk[pos]);                                    // inverse = conj(FFT(conj(.)))
        }
        buf[c * stride + i1] = v;
    }
    __syncthreads();
    lds_fft<TILE>(buf, tw, g.l1, stride);
    for (int e = tid; e < TILE; e += kFBlock) {
        const int c = e & (CW - 1), k1 = e >> lcw;
        const int64_t i2 = col0 + c;
        const cplx v = buf[c * stride + k1];
        if (MODE != 2) {
            wk[((int64_t)k1 << g.l2) + i2] = cmul(v, twiddle(i2 * k1, inv_n));
        } else {
            // natural order: k1 is the row i1 of the time-domain block
            const int64_t pos = ((int64_t)k1 << g.l2) + i2;
            if (pos < g.L - 1) continue;                           // the wrapped-around part of overlap-save
            const int64_t o0 = 2 * p * g.V + pos - (g.L - 1);
            if (o0 < g.n) out[o0 * g.out_ch + ch] = (float)(v.x * inv_n);
            const int64_t o1 = o0 + g.V;
            if (2 * p + 1 < g.nblocks && o1 < g.n) out[o1 * g.out_ch + ch] = (float)(-v.y * inv_n);   // conj
        }
    }
}

// FULL = false: forward row FFTs only (filter spectrum).  FULL = true: forward, times H, inverse,
// conjugate twiddle.  One workgroup = TILE / N2 consecutive rows = TILE consecutive points.
template <bool FULL, int TILE>
__global__ void __launch_bounds__(kFBlock)
k_fft_rows(cplx *work, ConvGeom g, const cplx *H, int fir_ch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int N2 = (int)g.N2;
    cplx *buf = reinterpret_cast<cplx *>(smem);
    cplx *tw = buf + TILE;
    const int tid = threadIdx.x;
    const int64_t pair = blockIdx.y;
    const int64_t row0 = ((int64_t)blockIdx.x * TILE) >> g.l2;
    cplx *wk = work + pair * g.N + (int64_t)blockIdx.x * TILE;
    const double inv_n = 1.0 / (double)g.N;
    fill_twiddles(tw, N2);
    for (int e = tid; e < TILE; e += kFBlock) buf[e] = wk[e];
    __syncthreads();
    lds_fft<TILE>(buf, tw, g.l2, N2);
    if (!FULL) {
        for (int e = tid; e < TILE; e += kFBlock) wk[e] = buf[e];
        return;
    }
    const int ch = (int)(pair / g.npairs);
    const cplx *Hc = H + (int64_t)(fir_ch == 1 ? 0 : ch) * g.N + (int64_t)blockIdx.x * TILE;
    for (int e = tid; e < TILE; e += kFBlock) buf[e] = cconj(cmul(buf[e], Hc[e]));
    __syncthreads();
    lds_fft<TILE>(buf, tw, g.l2, N2);
    for (int e = tid; e < TILE; e += kFBlock) {
        const int r = e >> g.l2, i2 = e & (N2 - 1);
        const int64_t k1 = row0 + r;
        // conj() completes the inverse row transform; the conjugate twiddle undoes step (1)'s
        wk[e] = cmul(cconj(buf[e]), cconj(twiddle(k1 * i2, inv_n)));
    }
}

// new history = the last L-1 samples of (history | x), per output channel.  dst == hist is allowed when
// n >= L-1 (every value then comes from x); shorter blocks go through a scratch copy.
__global__ void __launch_bounds__(kFBlock)
k_fft_hist(float *dst, const float *hist, const float *x, ConvGeom g) {
    const int64_t total = (g.L - 1) * g.out_ch;
    const int64_t stride = (int64_t)gridDim.x * kFBlock;
    for (int64_t e = (int64_t)blockIdx.x * kFBlock + threadIdx.x; e < total; e += stride) {
        const int64_t j = e / g.out_ch;
        const int c = (int)(e - j * g.out_ch);
        const int64_t k = g.n + j;                                 // index into (history | x)
        dst[e] = (k < g.L - 1) ? hist[k * g.out_ch + c]
                               : x[(k - (g.L - 1)) * g.src_ch + (g.src_ch == 1 ? 0 : c)];
    }
}

int fft_tile(int64_t fft_size) { return fft_size >= (1 << 18) ? 2048 : 1024; }

bool fft_geometry(int64_t fft_size, int64_t L, ConvGeom &g) {
    if (fft_size < 4096 || fft_size > (1 << 18) || (fft_size & (fft_size - 1))) return false;
    if (fft_size - (L - 1) < 1) return false;
    int lg = 0;
    while (((int64_t)1 << lg) < fft_size) ++lg;
    const int tile = fft_tile(fft_size);
    int l2 = 0;
    while ((1 << l2) < tile) ++l2;
    if (lg - l2 < 6) l2 = lg - 6;                  // N1 >= 64 (N2 = 64 for N = 4096)
    g.N = fft_size;
    g.l2 = l2;
    g.l1 = lg - l2;
    g.N2 = (int64_t)1 << g.l2;
    g.N1 = (int64_t)1 << g.l1;
    g.L = L;
    g.V = fft_size - (L - 1);
    return true;
}

size_t cols_smem(const ConvGeom &g, int tile) { return ((tile / g.N1) * (g.N1 + 1) + g.N1) * sizeof(cplx); }
size_t rows_smem(const ConvGeom &g, int tile) { return (tile + g.N2) * sizeof(cplx); }

template <int TILE>
int launch_prepare(cplx *H, const ConvGeom &g, const float *h, int fir_channels) {
    const dim3 grid((unsigned)(g.N / TILE), (unsigned)fir_channels);
    hipLaunchKernelGGL((k_fft_cols<1, TILE>), grid, dim3(kFBlock), cols_smem(g, TILE), pgx::stream(), H, g,
                       (const float *)nullptr, (const float *)nullptr, h, fir_channels, (float *)nullptr);
    PGX_LAUNCH_CHECK("k_fft_cols<filter>");
    hipLaunchKernelGGL((k_fft_rows<false, TILE>), grid, dim3(kFBlock), rows_smem(g, TILE), pgx::stream(), H, g,
                       (const cplx *)nullptr, fir_channels);
    PGX_LAUNCH_CHECK("k_fft_rows<filter>");
    return PGX_OK;
}

template <int TILE>
int launch_convolve(float *out, const float *x, const cplx *H, float *hist, cplx *work, const ConvGeom &g,
                    int fir_channels, int64_t pairs) {
    hipStream_t st = pgx::stream();
    const dim3 grid((unsigned)(g.N / TILE), (unsigned)pairs);
    hipLaunchKernelGGL((k_fft_cols<0, TILE>), grid, dim3(kFBlock), cols_smem(g, TILE), st, work, g, x,
                       (const float *)hist, (const float *)nullptr, fir_channels, (float *)nullptr);
    PGX_LAUNCH_CHECK("k_fft_cols<forward>");
    hipLaunchKernelGGL((k_fft_rows<true, TILE>), grid, dim3(kFBlock), rows_smem(g, TILE), st, work, g, H,
                       fir_channels);
    PGX_LAUNCH_CHECK("k_fft_rows");
    hipLaunchKernelGGL((k_fft_cols<2, TILE>), grid, dim3(kFBlock), cols_smem(g, TILE), st, work, g,
                       (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, fir_channels, out);
    PGX_LAUNCH_CHECK("k_fft_cols<inverse>");
    return PGX_OK;
}

}  // namespace

extern "C" {

int64_t pgx_convolve_fft_size(int64_t fir_len) {
    if (fir_len < 1) return 0;
    int64_t n = 4096;
    while (n < 2 * fir_len) n <<= 1;
    return n <= (1 << 18) ? n : 0;
}

size_t pgx_convolve_fft_spectrum_bytes(int64_t fft_size, int fir_channels) {
    if (fft_size <= 0 || fir_channels <= 0) return 0;
    return (size_t)fft_size * fir_channels * sizeof(cplx);
}

size_t pgx_convolve_fft_workspace_bytes(int64_t n, int64_t fir_len, int out_channels, int64_t fft_size) {
    ConvGeom g{};
    if (n <= 0 || out_channels <= 0 || !fft_geometry(fft_size, fir_len, g)) return 0;
    const int64_t nblocks = pgx::ceil_div(n, g.V);
    const int64_t npairs = (nblocks + 1) / 2;
    return (size_t)npairs * out_channels * fft_size * sizeof(cplx) + (size_t)(fir_len - 1) * out_channels * 4 + 64;
}

int pgx_convolve_fft_prepare(void *spectrum, const float *h, int64_t fir_len, int fir_channels, int64_t fft_size) {
    PGX_REQUIRE_INIT();
    ConvGeom g{};
    PGX_CHECK_ARG(spectrum && h && fir_len >= 1 && fir_channels >= 1, "pgx_convolve_fft_prepare: bad argument");
    PGX_CHECK_ARG(fft_geometry(fft_size, fir_len, g), "pgx_convolve_fft_prepare: unsupported fft size");
    g.n = 0; g.nblocks = 0; g.npairs = 1; g.src_ch = 1; g.out_ch = fir_channels;
    cplx *H = (cplx *)spectrum;
    return fft_tile(fft_size) == 2048 ? launch_prepare<2048>(H, g, h, fir_channels)
                                      : launch_prepare<1024>(H, g, h, fir_channels);
}

int pgx_convolve_fft(float *out, const float *x, int64_t n, int src_channels, const void *spectrum,
                     int64_t fir_len, int fir_channels, int out_channels, int64_t fft_size, float *hist,
                     void *workspace) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    ConvGeom g{};
    PGX_CHECK_ARG(out && x && spectrum && hist && workspace, "pgx_convolve_fft: null pointer");
    PGX_CHECK_ARG(fir_len >= 2 && src_channels >= 1 && fir_channels >= 1 && out_channels >= 1,
                  "pgx_convolve_fft: bad argument");
    PGX_CHECK_ARG(fft_geometry(fft_size, fir_len, g), "pgx_convolve_fft: unsupported fft size");
    PGX_CHECK_ARG((src_channels == 1 || src_channels == out_channels) &&
                  (fir_channels == 1 || fir_channels == out_channels), "pgx_convolve_fft: channel mismatch");
    g.n = n;
    g.nblocks = pgx::ceil_div(n, g.V);
    g.npairs = (g.nblocks + 1) / 2;
    g.src_ch = src_channels;
    g.out_ch = out_channels;
    const int64_t pairs = g.npairs * out_channels;
    PGX_CHECK_ARG(pairs <= 65535, "pgx_convolve_fft: block too long for one call");
    cplx *work = (cplx *)workspace;
    float *hist_new = (float *)(work + pairs * g.N);
    const cplx *H = (const cplx *)spectrum;
    hipStream_t st = pgx::stream();
    const int rc = fft_tile(fft_size) == 2048 ? launch_convolve<2048>(out, x, H, hist, work, g, fir_channels, pairs)
                                              : launch_convolve<1024>(out, x, H, hist, work, g, fir_channels, pairs);
    if (rc != PGX_OK) return rc;
    const int64_t hist_elems = (fir_len - 1) * out_channels;
    const bool in_place = n >= fir_len - 1;                     // then nothing of the old history is read
    hipLaunchKernelGGL(k_fft_hist, dim3(pgx::grid_for(hist_elems, kFBlock)), dim3(kFBlock), 0, st,
                       in_place ? hist : hist_new, (const float *)hist, x, g);
    PGX_LAUNCH_CHECK("k_fft_hist");
    if (!in_place) PGX_HIP(hipMemcpyAsync(hist, hist_new, hist_elems * sizeof(float), hipMemcpyDeviceToDevice, st));
    return PGX_OK;
}

}  // extern "C"
