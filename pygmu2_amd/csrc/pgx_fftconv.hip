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

#ifndef PGX_FFT_STAMP
#define PGX_FFT_STAMP(k, i)        /* tools/microbench/fft_phases.hip defines it to record wall_clock64() */
#endif

namespace {

constexpr int kFBlock = 256;
constexpr double kTwoPi = 6.283185307179586;

// 16-byte aligned: one ds_read_b128 / ds_write_b128 per element (consecutive lanes: conflict-free), where two
// 8-byte accesses at a 16-byte lane stride run into each other's banks
struct alignas(16) cplx {
    double x, y;
};
// (explicit FMAs: the transforms are not bound to a reference operation order, only to its float64 accuracy --
// one rounding less per product, and a third fewer instructions in a pass that is issue-bound)
__device__ __forceinline__ cplx cmul(const cplx &a, const cplx &b) {
    return cplx{__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x)};
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

// Per-stage twiddle tables in LDS.  A radix-4 stage with sub-transform length Ns needs W_{4Ns}^(k*t), k < Ns,
// t = 1..3; read from the one table W_M^p they sit (k*t) << (lm-2-lns) apart -- up to 16 lanes on the same
// banks.  Laid out per stage as [t][k] the lanes of a wave read consecutive words:
//   radix-4 stage Ns:  st[(Ns - 1) + (t - 1) * Ns + k]          (the stages before it hold 4^s - 1 = Ns - 1 entries)
//   final radix-2 stage (odd lm), Ns = M/2:  st[(Ns - 1) + k] = W_M^k
// M - 1 entries in all.  Filled from the global table g[p] = W_M^p.
// (PER = entries per thread, M <= 256 * PER.  All loads first -- unconditional, from a clamped entry -- then the LDS
// stores: a load inside the entry loop gets its s_waitcnt right behind it, the table then costs one memory latency per
// entry and, the counter being in order, drags the tile / spectrum / twiddle loads issued before it along.)
template <int PER>
__device__ __forceinline__ void fill_stage_twiddles(cplx *st, const cplx *g, int lm) {
    const int M = 1 << lm;
    cplx val[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        int e = threadIdx.x + k * kFBlock;
        if (e > M - 2) e = M - 2;
        // stage of entry e: the largest Ns = 4^s with Ns - 1 <= e
        int lns = (31 - __builtin_clz(e + 1)) & ~1;
        if (lns > lm - 1) lns = lm - 1;                           // (cannot happen for e < M - 1; keeps lns in range)
        const int Ns = 1 << lns, r = e - (Ns - 1);
        int p;
        if (lm - lns >= 2) {
            const int t = r >> lns, k2 = r & (Ns - 1);            // t = 0..2 -> W^(k*(t+1))
            p = (k2 * (t + 1)) << (lm - 2 - lns);
        } else {
            p = r;                                                // radix-2: W_M^k
        }
        val[k] = g[p];
    }
    // (the stores are unconditional too -- surplus threads write the spare entry M - 1 of the table's M slots -- or the
    // compiler sinks every load into its store's branch and waits for it there)
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int e = threadIdx.x + k * kFBlock;
        st[e < M - 1 ? e : M - 1] = val[k];
    }
}

// Forward DFT of TILE/M sequences of length M = 2^lm stored at a[s*stride + i], natural order in and out.
// Stockham autosort between two LDS images (`a` holds the input, `b` is scratch): one barrier per stage.
// `st`: the stage tables above.  All 256 threads take part; the input must be visible (barrier) on entry;
// returns the image that holds the result, visible to every thread.
// Stages with sub-transform length 2^lns_from <= Ns < 2^lns_to only (lns_from even): `a` then holds the output
// of the stages before them.
template <int TILE>
__device__ cplx *lds_fft_stages(cplx *a, cplx *b, const cplx *st, int lm, int stride, int lns_from, int lns_to) {
    const int tid = threadIdx.x;
    constexpr int U4 = TILE / 4 / kFBlock;                     // radix-4 butterflies per thread
    constexpr int U2 = TILE / 2 / kFBlock;                     // radix-2 butterflies per thread
    cplx *src = a, *dst = b;
    for (int lns = lns_from; lns < lns_to;) {
        const int Ns = 1 << lns;
        const cplx *tws = st + (Ns - 1);
        if (lm - lns >= 2) {
            const int lq = lm - 2, q = 1 << lq;                    // butterflies per sequence
#pragma unroll
            for (int u = 0; u < U4; ++u) {
                const int w = tid + u * kFBlock;
                const int s = w >> lq, j = w & (q - 1);
                const int k = j & (Ns - 1);
                const cplx *bi = src + s * stride;
                const cplx x0 = bi[j];
                const cplx c1 = cmul(bi[j + q], tws[k]);
                const cplx c2 = cmul(bi[j + 2 * q], tws[Ns + k]);
                const cplx c3 = cmul(bi[j + 3 * q], tws[2 * Ns + k]);
                const cplx s0 = cadd(x0, c2), s1 = csub(x0, c2), s2 = cadd(c1, c3), s3 = mul_neg_i(csub(c1, c3));
                cplx *bo = dst + s * stride + (j - k) * 4 + k;
                bo[0] = cadd(s0, s2);
                bo[Ns] = cadd(s1, s3);
                bo[2 * Ns] = csub(s0, s2);
                bo[3 * Ns] = csub(s1, s3);
            }
            lns += 2;
        } else {
            const int lq = lm - 1, q = 1 << lq;
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int w = tid + u * kFBlock;
                const int s = w >> lq, j = w & (q - 1);
                const int k = j & (Ns - 1);
                const cplx *bi = src + s * stride;
                const cplx x0 = bi[j];
                const cplx c1 = cmul(bi[j + q], tws[k]);
                cplx *bo = dst + s * stride + (j - k) * 2 + k;
                bo[0] = cadd(x0, c1);
                bo[Ns] = csub(x0, c1);
            }
            lns += 1;
        }
        __syncthreads();
        cplx *t = src;
        src = dst;
        dst = t;
    }
    return src;
}

template <int TILE>
__device__ cplx *lds_fft(cplx *a, cplx *b, const cplx *st, int lm, int stride) {
    return lds_fft_stages<TILE>(a, b, st, lm, stride, 0, lm);
}

// The first Stockham stage (radix 4, Ns = 1: no twiddles) on values a thread already holds: x[j + t*q], t = 0..3,
// q = M/4, come in, the four outputs go to positions 4j + t.
__device__ __forceinline__ void first_stage(const cplx &x0, const cplx &x1, const cplx &x2, const cplx &x3, cplx *at4j) {
    const cplx s0 = cadd(x0, x2), s1 = csub(x0, x2), s2 = cadd(x1, x3), s3 = mul_neg_i(csub(x1, x3));
    at4j[0] = cadd(s0, s2);
    at4j[1] = cadd(s1, s3);
    at4j[2] = csub(s0, s2);
    at4j[3] = csub(s1, s3);
}

// The stage twiddles of a thread.  A thread's butterfly of a middle stage (sub-transform length Ns) is w = tid + u*256,
// position j = w & (M/4 - 1), twiddle index k = j & (Ns - 1): the same three factors W_4Ns^(k*t) in every transform
// the workgroup runs, and so are those of the last stage (the thread's own outputs).  They are gathered once from the
// global table g[p] = W_M^p into registers: no table in LDS (16 KB less per workgroup: five workgroups per CU
// instead of three), and a radix-4 butterfly reads four LDS words instead of seven.
template <int TILE, int LM>
struct StageTwiddles {
    static constexpr int PT = TILE / kFBlock, U4 = PT / 4, U2 = PT / 2;
    static constexpr int LAST = (LM & 1) ? LM - 1 : LM - 2;       // lns of the final stage
    static constexpr int MID = (LAST - 2) / 2;                    // stages lns = 2, 4 .. LAST - 2 go through LDS
    static constexpr int NLAST = (LM & 1) ? U2 : 3 * U4;
    cplx mid[MID > 0 ? MID : 1][U4][3];
    cplx last[NLAST];

    // j0, jstep: the thread consumes X[j0 + u*jstep] (see tile_fft_regs)
    __device__ __forceinline__ void load(const cplx *g, int j0, int jstep) {
#pragma unroll
        for (int m = 0; m < MID; ++m) {
            const int lns = 2 + 2 * m, Ns = 1 << lns;
#pragma unroll
            for (int u = 0; u < U4; ++u) {
                const int w = threadIdx.x + u * kFBlock;
                const int k = w & ((1 << (LM - 2)) - 1) & (Ns - 1);
#pragma unroll
                for (int t = 1; t <= 3; ++t) mid[m][u][t - 1] = g[(k * t) << (LM - 2 - lns)];
            }
        }
        if constexpr (LM & 1) {
#pragma unroll
            for (int bf = 0; bf < U2; ++bf) last[bf] = g[j0 + bf * jstep];
        } else {
#pragma unroll
            for (int bf = 0; bf < U4; ++bf) {
                const int j = j0 + bf * jstep;
#pragma unroll
                for (int t = 1; t <= 3; ++t) last[3 * bf + t - 1] = g[j * t];
            }
        }
    }
};

// DFT of the sequences (length M = 2^LM, `stride` apart) of a tile with the first and the last stage in registers:
// those are the butterflies whose operands / results are the thread's own elements x[j0 + u*jstep] (u < PT,
// jstep = M/PT) of the sequence at `seq_off`, so of the LM/2 LDS round trips (write, barrier, read) two disappear.
// `a`, `b`: the two LDS images (contents irrelevant; nobody may still be reading `a`).  Returns the image last read
// (the other one is free to write), with X[j0 + u*jstep] in v[u].
template <int TILE, int LM>
__device__ __forceinline__ cplx *tile_fft_regs(cplx (&v)[TILE / kFBlock], cplx *a, cplx *b,
                                               const StageTwiddles<TILE, LM> &tw, int stride, int seq_off, int j0,
                                               int jstep) {
    using TW = StageTwiddles<TILE, LM>;
    constexpr int PT = TW::PT, U4 = TW::U4, U2 = TW::U2;
    const int tid = threadIdx.x;
#pragma unroll
    for (int bf = 0; bf < U4; ++bf)
        first_stage(v[bf], v[bf + U4], v[bf + 2 * U4], v[bf + 3 * U4], a + seq_off + 4 * (j0 + bf * jstep));
    __syncthreads();
    cplx *src = a, *dst = b;
    constexpr int lq = LM - 2, q = 1 << lq;                      // radix-4 butterflies per sequence
#pragma unroll
    for (int m = 0; m < TW::MID; ++m) {
        const int Ns = 1 << (2 + 2 * m);
#pragma unroll
        for (int u = 0; u < U4; ++u) {
            const int w = tid + u * kFBlock;
            const int s = w >> lq, j = w & (q - 1);
            const int k = j & (Ns - 1);
            const cplx *bi = src + s * stride;
            const cplx x0 = bi[j];
            const cplx c1 = cmul(bi[j + q], tw.mid[m][u][0]);
            const cplx c2 = cmul(bi[j + 2 * q], tw.mid[m][u][1]);
            const cplx c3 = cmul(bi[j + 3 * q], tw.mid[m][u][2]);
            const cplx s0 = cadd(x0, c2), s1 = csub(x0, c2), s2 = cadd(c1, c3), s3 = mul_neg_i(csub(c1, c3));
            cplx *bo = dst + s * stride + (j - k) * 4 + k;
            bo[0] = cadd(s0, s2);
            bo[Ns] = cadd(s1, s3);
            bo[2 * Ns] = csub(s0, s2);
            bo[3 * Ns] = csub(s1, s3);
        }
        __syncthreads();
        cplx *t = src;
        src = dst;
        dst = t;
    }
    // the last stage (Ns = q or M/2: k = j) into registers: exactly the butterflies whose outputs are the thread's
    const cplx *seq = src + seq_off;
    constexpr int Ns = 1 << TW::LAST;
    if constexpr (LM & 1) {                                     // radix 2: outputs j, j + M/2
#pragma unroll
        for (int bf = 0; bf < U2; ++bf) {
            const int j = j0 + bf * jstep;
            const cplx x0 = seq[j];
            const cplx c1 = cmul(seq[j + Ns], tw.last[bf]);
            v[bf] = cadd(x0, c1);
            v[bf + U2] = csub(x0, c1);
        }
    } else {                                                    // radix 4: outputs j + t * M/4
#pragma unroll
        for (int bf = 0; bf < U4; ++bf) {
            const int j = j0 + bf * jstep;
            const cplx x0 = seq[j];
            const cplx c1 = cmul(seq[j + Ns], tw.last[3 * bf]);
            const cplx c2 = cmul(seq[j + 2 * Ns], tw.last[3 * bf + 1]);
            const cplx c3 = cmul(seq[j + 3 * Ns], tw.last[3 * bf + 2]);
            const cplx s0 = cadd(x0, c2), s1 = csub(x0, c2), s2 = cadd(c1, c3), s3 = mul_neg_i(csub(c1, c3));
            v[bf] = cadd(s0, s2);
            v[bf + U4] = cadd(s1, s3);
            v[bf + 2 * U4] = csub(s0, s2);
            v[bf + 3 * U4] = csub(s1, s3);
        }
    }
    return src;
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
    int hist_zero;        // the overlap history is all zeros (fresh stream): its buffer is not read
    int mixed;            // one filter for every channel: any two (channel, block) items may share a transform
    int stereo;           // mixed, two interleaved channels in and out: a transform = (left, right) of one block
};

// Twiddle tables, made once per filter next to its spectrum (pgx_convolve_fft_prepare):
//   big[(k1 << l2) + i2] = W_N^(i2*k1)   the four-step twiddle, in the layout of the work buffer
//   t1[p] = W_N1^p,  t2[p] = W_N2^p      the in-LDS FFT twiddles
struct Tables {
    const cplx *big, *t1, *t2;
};

__global__ void __launch_bounds__(kFBlock)
k_fft_tables(cplx *big, cplx *t1, cplx *t2, ConvGeom g) {
    const double inv_n = 1.0 / (double)g.N;
    const int64_t stride = (int64_t)gridDim.x * kFBlock;
    for (int64_t e = (int64_t)blockIdx.x * kFBlock + threadIdx.x; e < g.N; e += stride) {
        const int64_t k1 = e >> g.l2, i2 = e & (g.N2 - 1);
        big[e] = twiddle(i2 * k1, inv_n);
        if (e < g.N1) t1[e] = twiddle(e, 1.0 / (double)g.N1);
        if (e < g.N2) t2[e] = twiddle(e, 1.0 / (double)g.N2);
    }
}

// sample `pos` of overlap-save block `b` of output channel `ch`: (history | x) at b*V + pos.
// Both candidates are loaded unconditionally (from index 0 when the sample is not theirs) and selected afterwards:
// a load behind a branch gets its s_waitcnt at the join, i.e. immediately, and a thread's sixteen loads then run one
// memory latency after the other (ISA of the first version: global_load_dword / s_waitcnt vmcnt(0) pairs).
__device__ __forceinline__ double conv_input(const ConvGeom &g, const float *x, const float *hist, int64_t b, int ch,
                                             int64_t pos) {
    const int64_t e = b * g.V + pos;                               // index into (history | x)
    const bool block_ok = b < g.nblocks;
    const bool in_hist = e < g.L - 1;
    const int64_t i = e - (g.L - 1);
    const bool from_hist = block_ok && in_hist && !g.hist_zero;
    const bool from_x = block_ok && !in_hist && i < g.n;
    const unsigned hv = __float_as_uint(hist[from_hist ? e * g.out_ch + ch : 0]);
    const unsigned xv = __float_as_uint(x[from_x ? i * g.src_ch + (g.src_ch == 1 ? 0 : ch) : 0]);
    // (bit masks, not ?: -- the compiler turns selects of loaded values back into branches around the loads)
    const unsigned bits = (hv & (from_hist ? 0xffffffffu : 0u)) | (xv & (from_x ? 0xffffffffu : 0u));
    return (double)__uint_as_float(bits);
}

// Stereo pairs: frame (left, right) of block b as one complex sample -- one 8-byte load where the general form
// needs two 4-byte ones from two places (and one 8-byte store on the way out).
__device__ __forceinline__ cplx conv_input_stereo(const ConvGeom &g, const float2 *x, const float2 *hist, int64_t b,
                                                  int64_t pos) {
    const int64_t e = b * g.V + pos;
    const bool in_hist = e < g.L - 1;
    const int64_t i = e - (g.L - 1);
    const bool from_hist = in_hist && !g.hist_zero;
    const bool from_x = !in_hist && i < g.n;
    // one load from a selected address (x[0] when the sample is nobody's), masked afterwards
    const float2 *src = from_hist ? hist + e : x + (from_x ? i : 0);
    const float2 val = *src;
    const unsigned m = (from_hist || from_x) ? 0xffffffffu : 0u;
    return cplx{(double)__uint_as_float(__float_as_uint(val.x) & m), (double)__uint_as_float(__float_as_uint(val.y) & m)};
}

// The two real sequences packed into transform `pair` as real and imaginary part.  The filter is real, so they
// never mix.  With one filter for all channels the items (channel, block), numbered ch*nblocks + b, are paired as
// they come -- a single 65 537-frame stereo block is one transform, not two half-empty ones; with a filter per
// channel a transform holds blocks 2p and 2p+1 of one channel.  b >= nblocks: an empty slot.
struct PairItems {
    int ch0, ch1;
    int64_t b0, b1;
};
__device__ __forceinline__ PairItems pair_items(const ConvGeom &g, int64_t pair) {
    PairItems it;
    if (g.stereo) {
        it.ch0 = 0;
        it.ch1 = 1;
        it.b0 = it.b1 = pair;
    } else if (g.mixed) {
        // (items and pairs fit 32 bits -- pairs <= 65535 per call -- and 64-bit divisions are long loops)
        const unsigned nb = (unsigned)g.nblocks, q0 = 2u * (unsigned)pair, q1 = q0 + 1u;
        it.ch0 = (int)(q0 / nb);
        it.b0 = q0 - (unsigned)it.ch0 * nb;
        it.ch1 = (int)(q1 / nb);
        it.b1 = q1 - (unsigned)it.ch1 * nb;
        if (it.ch1 >= g.out_ch) {
            it.ch1 = it.ch0;
            it.b1 = g.nblocks;
        }
    } else {
        const unsigned np = (unsigned)g.npairs;
        it.ch0 = it.ch1 = (int)((unsigned)pair / np);
        it.b0 = 2u * ((unsigned)pair - (unsigned)it.ch0 * np);
        it.b1 = it.b0 + 1;
    }
    return it;
}

// MODE 0: forward, input = packed signal blocks; MODE 1: forward, input = filter taps (spectrum
// preparation); MODE 2: inverse, output = float32 samples.
// (L1 = log2 N1, known at compile time: 6 up to N = 2^16, 7 above)
template <int MODE, int TILE, int L1>
__global__ void __launch_bounds__(kFBlock)
k_fft_cols(cplx *work, ConvGeom g, Tables tb, const float *x, const float *hist, const float *h, int fir_ch,
           float *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PT = TILE / kFBlock;                             // points per thread
    constexpr int N1 = 1 << L1;
    constexpr int CW = TILE >> L1, lcw = __builtin_ctz(CW);        // columns per workgroup
    constexpr int stride = N1 + 1;                                 // +1: spread the columns over the banks
    cplx *buf = reinterpret_cast<cplx *>(smem);
    cplx *alt = buf + CW * stride;
    const int tid = threadIdx.x;
    const int64_t pair = blockIdx.y;
    const int64_t col0 = (int64_t)blockIdx.x * CW;
    cplx *wk = work + pair * g.N;
    const int ch = (int)pair;                                      // MODE 1: the filter channel
    PairItems it{};
    if (MODE != 1) it = pair_items(g, pair);
    const double inv_n = 1.0 / (double)g.N;
    PGX_FFT_STAMP(MODE, 0);

    // everything that comes from HBM is requested first.  The twiddles go in front of the inputs: the input loads sit
    // in a branch (stereo / general form) whose float -> double conversions the compiler keeps there, with a wait for
    // everything issued so far -- issued behind it, the twiddles were a second memory round trip.
    // The thread's elements are column tid & (CW-1), rows (tid >> lcw) + u * N1/PT -- in the time domain and, in
    // the same registers, in the frequency domain
    StageTwiddles<TILE, L1> tw;
    tw.load(tb.t1, tid >> lcw, N1 / PT);
    cplx bigtw[MODE != 2 ? PT : 1];
    if (MODE != 2) {
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int e = tid + u * kFBlock;
            const int c = e & (CW - 1), k1 = e >> lcw;
            bigtw[u] = tb.big[((int64_t)k1 << g.l2) + col0 + c];
        }
    }
    cplx v[PT];
    if (MODE == 0 && g.stereo) {
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int e = tid + u * kFBlock;
            const int c = e & (CW - 1), i1 = e >> lcw;
            v[u] = conv_input_stereo(g, reinterpret_cast<const float2 *>(x), reinterpret_cast<const float2 *>(hist),
                                     it.b0, ((int64_t)i1 << g.l2) + col0 + c);
        }
    } else {
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int e = tid + u * kFBlock;
            const int c = e & (CW - 1), i1 = e >> lcw;
            const int64_t pos = ((int64_t)i1 << g.l2) + col0 + c;
            if (MODE == 0) {
                v[u] = cplx{conv_input(g, x, hist, it.b0, it.ch0, pos), conv_input(g, x, hist, it.b1, it.ch1, pos)};
            } else if (MODE == 1) {
                v[u] = cplx{pos < g.L ? (double)h[pos * fir_ch + ch] : 0.0, 0.0};
            } else {
                v[u] = cconj(wk[pos]);                             // inverse = conj(FFT(conj(.)))
            }
        }
    }
    PGX_FFT_STAMP(MODE, 1);
    tile_fft_regs<TILE, L1>(v, buf, alt, tw, stride, (tid & (CW - 1)) * stride, tid >> lcw, N1 / PT);
    PGX_FFT_STAMP(MODE, 2);
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int e = tid + u * kFBlock;
        const int c = e & (CW - 1), k1 = e >> lcw;
        const int64_t i2 = col0 + c;
        const cplx r = v[u];
        if (MODE != 2) {
            wk[((int64_t)k1 << g.l2) + i2] = cmul(r, bigtw[u]);
        } else {
            // natural order: k1 is the row i1 of the time-domain block
            const int64_t pos = ((int64_t)k1 << g.l2) + i2;
            if (pos < g.L - 1) continue;                           // the wrapped-around part of overlap-save
            if (g.stereo) {
                const int64_t o = it.b0 * g.V + pos - (g.L - 1);
                if (o < g.n) reinterpret_cast<float2 *>(out)[o] = make_float2((float)(r.x * inv_n), (float)(-r.y * inv_n));
                continue;
            }
            const int64_t o0 = it.b0 * g.V + pos - (g.L - 1);
            if (o0 < g.n) out[o0 * g.out_ch + it.ch0] = (float)(r.x * inv_n);
            const int64_t o1 = it.b1 * g.V + pos - (g.L - 1);
            if (it.b1 < g.nblocks && o1 < g.n) out[o1 * g.out_ch + it.ch1] = (float)(-r.y * inv_n);   // conj
        }
    }
    PGX_FFT_STAMP(MODE, 3);
}

// FULL = false: forward row FFTs only (filter spectrum).  FULL = true: forward, times H, inverse,
// conjugate twiddle.  One workgroup = TILE / N2 consecutive rows = TILE consecutive points.
// hist_dst != nullptr (FULL only, blocks of at least L-1 frames): the workgroups also move the overlap
// history on -- its new content is the last L-1 input frames, and the pass that read the old one is over.
template <bool FULL, int TILE>
__global__ void __launch_bounds__(kFBlock)
k_fft_rows(cplx *work, ConvGeom g, Tables tb, const cplx *H, int fir_ch, float *hist_dst, const float *x) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PT = TILE / kFBlock;
    const int N2 = (int)g.N2;
    constexpr int LT = __builtin_ctz(TILE);
    cplx *buf = reinterpret_cast<cplx *>(smem);
    cplx *alt = buf + TILE;
    const int tid = threadIdx.x;
    const int64_t pair = blockIdx.y;
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    cplx *wk = work + pair * g.N + tile0;
    cplx v[PT], hv[FULL ? PT : 1], bigtw[FULL ? PT : 1];
    PGX_FFT_STAMP(4, 0);
#pragma unroll
    for (int u = 0; u < PT; ++u) v[u] = wk[tid + u * kFBlock];
    if (FULL) {
        // (32-bit: a 64-bit division is a loop of a hundred instructions in front of the first transform)
        const unsigned ch = fir_ch == 1 ? 0u : (unsigned)blockIdx.y / (unsigned)g.npairs;
        const cplx *Hc = H + (int64_t)ch * g.N + tile0;
#pragma unroll
        for (int u = 0; u < PT; ++u) hv[u] = Hc[tid + u * kFBlock];
    }
    // the workgroup's share of the new history: requested now, stored at the very end (a copy loop after the
    // transforms was 0.75 us of every workgroup's life: one more memory round trip)
    constexpr int HP = 4;
    float hval[FULL ? HP : 1];
    unsigned hdst[FULL ? HP : 1];
    unsigned hist_next = 0, hist_end = 0;
    if (FULL) {
        const unsigned oc = (unsigned)g.out_ch;
        const unsigned total = hist_dst != nullptr ? (unsigned)(g.L - 1) * oc : 0u;
        const unsigned groups = gridDim.x * gridDim.y, me = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned share = (total + groups - 1) / groups;
        hist_end = (me + 1) * share < total ? (me + 1) * share : total;
#pragma unroll
        for (int u = 0; u < HP; ++u) {
            const unsigned e = me * share + tid + u * kFBlock;
            const bool mine = e < hist_end;
            const unsigned j = e / oc, c = e - j * oc;
            hval[u] = x[mine ? (g.n + j - (g.L - 1)) * g.src_ch + (g.src_ch == 1 ? 0 : c) : 0];
            hdst[u] = mine ? e : ~0u;
        }
        hist_next = me * share + tid + HP * kFBlock;
    }
    if (FULL && N2 == TILE) {
        // one row per workgroup: first and last stage of both transforms in registers, and their twiddles too
        StageTwiddles<TILE, LT> tw;
        tw.load(tb.t2, tid, kFBlock);
        PGX_FFT_STAMP(4, 1);
        cplx *read_last = tile_fft_regs<TILE, LT>(v, buf, alt, tw, TILE, 0, tid, kFBlock);
        PGX_FFT_STAMP(4, 2);
        // (the conjugate twiddles of the way back are requested here, behind the spectrum that has just been used up:
        // sixteen registers fewer while both are alive, and the inverse transform covers the latency)
#pragma unroll
        for (int u = 0; u < PT; ++u) bigtw[FULL ? u : 0] = tb.big[tile0 + tid + u * kFBlock];
#pragma unroll
        for (int u = 0; u < PT; ++u) v[u] = cconj(cmul(v[u], hv[FULL ? u : 0]));
        tile_fft_regs<TILE, LT>(v, read_last == buf ? alt : buf, read_last, tw, TILE, 0, tid, kFBlock);
        PGX_FFT_STAMP(4, 3);
        // conj() completes the inverse row transform; the conjugate twiddle undoes step (1)'s
#pragma unroll
        for (int u = 0; u < PT; ++u) wk[tid + u * kFBlock] = cmul(cconj(v[u]), cconj(bigtw[FULL ? u : 0]));
    } else {
        cplx *tw = alt + TILE;                                      // short rows: stage tables in LDS
        if (FULL) {
#pragma unroll
            for (int u = 0; u < PT; ++u) bigtw[FULL ? u : 0] = tb.big[tile0 + tid + u * kFBlock];
        }
        fill_stage_twiddles<TILE / kFBlock>(tw, tb.t2, g.l2);       // N2 <= TILE
#pragma unroll
        for (int u = 0; u < PT; ++u) buf[tid + u * kFBlock] = v[u];
        __syncthreads();
        cplx *res = lds_fft<TILE>(buf, alt, tw, g.l2, N2);
        if (!FULL) {
#pragma unroll
            for (int u = 0; u < PT; ++u) wk[tid + u * kFBlock] = res[tid + u * kFBlock];
            return;
        }
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int e = tid + u * kFBlock;
            res[e] = cconj(cmul(res[e], hv[FULL ? u : 0]));       // own elements only: no barrier needed before
        }
        __syncthreads();
        const cplx *fin = lds_fft<TILE>(res, res == buf ? alt : buf, tw, g.l2, N2);
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int e = tid + u * kFBlock;
            wk[e] = cmul(cconj(fin[e]), cconj(bigtw[FULL ? u : 0]));
        }
    }
    PGX_FFT_STAMP(4, 4);
    if (FULL) {
#pragma unroll
        for (int u = 0; u < HP; ++u)
            if (hdst[u] != ~0u) hist_dst[hdst[u]] = hval[u];
        for (unsigned e = hist_next; e < hist_end; e += kFBlock) {      // (few workgroups, a long filter)
            const unsigned j = e / (unsigned)g.out_ch, c = e - j * (unsigned)g.out_ch;
            hist_dst[e] = x[(g.n + j - (g.L - 1)) * g.src_ch + (g.src_ch == 1 ? 0 : c)];
        }
    }
    PGX_FFT_STAMP(4, 5);
}

// new history = the last L-1 samples of (history | x), per output channel, for blocks shorter than L-1
// frames (longer ones: k_fft_rows).  Goes through a scratch copy: part of the old history survives.
__global__ void __launch_bounds__(kFBlock)
k_fft_hist(float *dst, const float *hist, const float *x, ConvGeom g) {
    const int64_t total = (g.L - 1) * g.out_ch;
    const int64_t stride = (int64_t)gridDim.x * kFBlock;
    for (int64_t e = (int64_t)blockIdx.x * kFBlock + threadIdx.x; e < total; e += stride) {
        const int64_t j = e / g.out_ch;
        const int c = (int)(e - j * g.out_ch);
        const int64_t k = g.n + j;                                 // index into (history | x)
        dst[e] = (k < g.L - 1) ? (g.hist_zero ? 0.0f : hist[k * g.out_ch + c])
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

size_t cols_smem(const ConvGeom &g, int tile) { return 2 * (tile / g.N1) * (g.N1 + 1) * sizeof(cplx); }
// (rows as long as the tile keep their stage twiddles in registers; shorter ones in an LDS table of N2 entries)
size_t rows_smem(const ConvGeom &g, int tile, bool full) {
    return (2 * tile + (full && g.N2 == tile ? 0 : g.N2)) * sizeof(cplx);
}

// the spectrum blob: [H: fir_channels x N][big: N][t1: N1][t2: N2] complex doubles
Tables tables_of(const cplx *spectrum, const ConvGeom &g, int fir_channels) {
    const cplx *big = spectrum + (int64_t)fir_channels * g.N;
    return Tables{big, big + g.N, big + g.N + g.N1};
}

// LDS images beyond the 64 KB a kernel gets by default (the 2^18-point geometry)
// (per kernel once, for the largest geometry: the call costs tens of microseconds of host time)
template <typename K>
int allow_lds(K kernel, size_t bytes) {
    static size_t allowed = 64 * 1024;
    if (bytes > allowed) {
        PGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)bytes));
        allowed = bytes;
    }
    return PGX_OK;
}

template <int TILE, int L1>
int launch_prepare(cplx *H, const ConvGeom &g, const float *h, int fir_channels) {
    const Tables tb = tables_of(H, g, fir_channels);
    hipLaunchKernelGGL(k_fft_tables, dim3(pgx::grid_for(g.N, kFBlock)), dim3(kFBlock), 0, pgx::stream(),
                       const_cast<cplx *>(tb.big), const_cast<cplx *>(tb.t1), const_cast<cplx *>(tb.t2), g);
    PGX_LAUNCH_CHECK("k_fft_tables");
    const dim3 grid((unsigned)(g.N / TILE), (unsigned)fir_channels);
    const size_t cols_lds = cols_smem(g, TILE), rows_lds = rows_smem(g, TILE, false);
    if (int rc = allow_lds(k_fft_cols<1, TILE, L1>, cols_lds)) return rc;
    if (int rc = allow_lds(k_fft_rows<false, TILE>, rows_lds)) return rc;
    hipLaunchKernelGGL((k_fft_cols<1, TILE, L1>), grid, dim3(kFBlock), cols_lds, pgx::stream(), H, g, tb,
                       (const float *)nullptr, (const float *)nullptr, h, fir_channels, (float *)nullptr);
    PGX_LAUNCH_CHECK("k_fft_cols<filter>");
    hipLaunchKernelGGL((k_fft_rows<false, TILE>), grid, dim3(kFBlock), rows_lds, pgx::stream(), H, g, tb,
                       (const cplx *)nullptr, fir_channels, (float *)nullptr, (const float *)nullptr);
    PGX_LAUNCH_CHECK("k_fft_rows<filter>");
    return PGX_OK;
}

struct ConvCall {
    float *out;
    const float *x;
    const cplx *H;
    float *hist, *hist_in_place;
    cplx *work;
    int fir_channels;
    int64_t pairs;
};

template <int TILE, int L1>
int launch_convolve(const ConvCall &c, const ConvGeom &g) {
    hipStream_t st = pgx::stream();
    const Tables tb = tables_of(c.H, g, c.fir_channels);
    const dim3 grid((unsigned)(g.N / TILE), (unsigned)c.pairs);
    const size_t cols_lds = cols_smem(g, TILE), rows_lds = rows_smem(g, TILE, true);
    if (int rc = allow_lds(k_fft_cols<0, TILE, L1>, cols_lds)) return rc;
    if (int rc = allow_lds(k_fft_cols<2, TILE, L1>, cols_lds)) return rc;
    if (int rc = allow_lds(k_fft_rows<true, TILE>, rows_lds)) return rc;
    hipLaunchKernelGGL((k_fft_cols<0, TILE, L1>), grid, dim3(kFBlock), cols_lds, st, c.work, g, tb, c.x,
                       (const float *)c.hist, (const float *)nullptr, c.fir_channels, (float *)nullptr);
    PGX_LAUNCH_CHECK("k_fft_cols<forward>");
    hipLaunchKernelGGL((k_fft_rows<true, TILE>), grid, dim3(kFBlock), rows_lds, st, c.work, g, tb, c.H,
                       c.fir_channels, c.hist_in_place, c.x);
    PGX_LAUNCH_CHECK("k_fft_rows");
    hipLaunchKernelGGL((k_fft_cols<2, TILE, L1>), grid, dim3(kFBlock), cols_lds, st, c.work, g, tb,
                       (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, c.fir_channels, c.out);
    PGX_LAUNCH_CHECK("k_fft_cols<inverse>");
    return PGX_OK;
}

// the three geometries fft_geometry() produces: (tile, log2 N1) = (2048, 7) for 2^18, (1024, 7) for 2^17, (1024, 6) below
#define PGX_FFT_BY_GEOMETRY(g, call)                                         \
    (fft_tile((g).N) == 2048 ? call<2048, 7> : (g).l1 == 7 ? call<1024, 7> : call<1024, 6>)

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
    ConvGeom g{};
    if (!fft_geometry(fft_size, 2, g)) return 0;
    return ((size_t)fft_size * fir_channels + (size_t)fft_size + g.N1 + g.N2) * sizeof(cplx);   // H + twiddle tables
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
    g.n = 0; g.nblocks = 0; g.npairs = 1; g.src_ch = 1; g.out_ch = fir_channels; g.hist_zero = 0; g.mixed = 0;
    g.stereo = 0;
    cplx *H = (cplx *)spectrum;
    return PGX_FFT_BY_GEOMETRY(g, launch_prepare)(H, g, h, fir_channels);
}

int pgx_convolve_fft(float *out, const float *x, int64_t n, int src_channels, const void *spectrum,
                     int64_t fir_len, int fir_channels, int out_channels, int64_t fft_size, float *hist,
                     void *workspace, int hist_is_zero) {
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
    g.hist_zero = hist_is_zero ? 1 : 0;
    g.mixed = fir_channels == 1 ? 1 : 0;
    g.stereo = g.mixed && src_channels == 2 && out_channels == 2 &&
               (((uintptr_t)x | (uintptr_t)out | (uintptr_t)hist) & 7) == 0;
    const int64_t pairs = g.mixed ? (g.nblocks * out_channels + 1) / 2 : g.npairs * out_channels;
    PGX_CHECK_ARG(pairs <= 65535, "pgx_convolve_fft: block too long for one call");
    cplx *work = (cplx *)workspace;
    float *hist_new = (float *)(work + pairs * g.N);
    const cplx *H = (const cplx *)spectrum;
    hipStream_t st = pgx::stream();
    const bool in_place = n >= fir_len - 1;                     // then nothing of the old history survives:
    float *hip = in_place ? hist : nullptr;                     // the row pass rewrites it on the side
    const ConvCall call{out, x, H, hist, hip, work, fir_channels, pairs};
    const int rc = PGX_FFT_BY_GEOMETRY(g, launch_convolve)(call, g);
    if (rc != PGX_OK) return rc;
    if (!in_place) {
        const int64_t hist_elems = (fir_len - 1) * out_channels;
        hipLaunchKernelGGL(k_fft_hist, dim3(pgx::grid_for(hist_elems, kFBlock)), dim3(kFBlock), 0, st, hist_new,
                           (const float *)hist, x, g);
        PGX_LAUNCH_CHECK("k_fft_hist");
        PGX_HIP(hipMemcpyAsync(hist, hist_new, hist_elems * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    return PGX_OK;
}

}  // extern "C"
