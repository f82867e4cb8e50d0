// pgx_scan.hip -- linear-recurrence PEs evaluated as parallel scans over affine state maps.
//
//   BiquadPE (constant)  : 2-state constant map      z' = A z + B x      (biquad_pe.py:383-404)
//   BiquadPE (varying)   : 2-state time-varying map  [y;y1]' = M_n [y1;y2] + [ff_n;0] (biquad_pe.py:35-62)
//   BlitSawPE            : prefix sum (phase) + 1-state constant map (leaky integrator)
//                          (blit_saw_pe.py:150-264)
//   SinePE (stateful)    : prefix sum (phase)        (sine_pe.py:177-232)
//   SVFilterPE           : 2-state map with a full 2x2 A, constant or per sample (svfilter_pe.py:41-205)
//   EnvelopePE           : 1-state map; with attack != release a walk over regimes (envelope_pe.py:128-271)
//   TransformPE          : element-wise op chains (transform_pe.py:96-152) -- no recurrence, kept here
//                          with the PEs of the same reference script
//
// Common structure (wave64, 256-thread workgroups = 4 waves):
//   * a tile is 256 threads x T consecutive frames; every thread folds its T frames into one
//     affine map starting from the zero state (pass 1),
//   * the 64 per-lane maps of a wave are combined with a Kogge-Stone scan over __shfl_up
//     (6 steps; for constant maps only the offset vector travels, the matrix powers
//     A^(T*2^k) are loop invariants kept in LDS),
//   * the 4 wave totals cross through LDS; the tile's carry-out feeds the next tile of the
//     same chain in registers,
//   * every thread then re-runs its T frames from its scanned carry-in in EXACTLY the
//     reference's float64 operation order (pass 2) and stores float32.
// So the only departure from the reference's sequential float64 arithmetic is the rounding
// of each chunk's carry-in (O(1e-16) relative); everything is compiled with -ffp-contract=off.
//
// Long chains are cut into segments, one workgroup each, in one of three ways:
//   * constant sections that forget (k_biquad_settled): each workgroup rebuilds its carry-in from a
//     short warm-up over the frames before its range -- one launch, no cross-workgroup traffic;
//   * constant sections that decay slowly (k_biquad_const<reduce/apply>): a reduce launch produces every
//     segment's zero-state response, the apply launch folds the preceding aggregates (binary powers of
//     A^segment) into its carry-in -- exact for any section;
//   * time-varying maps (k_biquad_varying / k_svf <reduce/apply>): the reduce launch composes each
//     segment's affine map, the apply launch folds the earlier maps onto the carried state.
// Many short chains (voices) use one workgroup per chain and no cross-workgroup traffic at all.

#include <cstdlib>
#include <type_traits>

#include "pgx_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

struct V2 {
    double x, y;
};
struct M2 {
    double a, b, c, d;   // [[a b],[c d]]
};

__device__ __forceinline__ M2 mm(const M2 &p, const M2 &q) {   // p*q
    M2 r;
    r.a = p.a * q.a + p.b * q.c;
    r.b = p.a * q.b + p.b * q.d;
    r.c = p.c * q.a + p.d * q.c;
    r.d = p.c * q.b + p.d * q.d;
    return r;
}
__device__ __forceinline__ V2 mv(const M2 &p, const V2 &v) {
    V2 r;
    r.x = p.a * v.x + p.b * v.y;
    r.y = p.c * v.x + p.d * v.y;
    return r;
}
__device__ __forceinline__ V2 vadd(const V2 &p, const V2 &q) { return V2{p.x + q.x, p.y + q.y}; }
__device__ __forceinline__ M2 m_identity() { return M2{1.0, 0.0, 0.0, 1.0}; }

__device__ __forceinline__ V2 shfl_up_v2(const V2 &v, int d) {
    return V2{__shfl_up(v.x, d, 64), __shfl_up(v.y, d, 64)};
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {   // src_lane wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64_keep(double old, double v) {     // lanes without a source keep `old`
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// DPP shifts of a 2x2 map / a 2-vector; lanes without a source see the identity map / the zero vector, so a
// scan step needs no lane test.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ M2 dpp_m2_identity(const M2 &m) {
    return M2{dpp_f64_keep<CTRL, ROW_MASK>(1.0, m.a), dpp_f64_keep<CTRL, ROW_MASK>(0.0, m.b),
              dpp_f64_keep<CTRL, ROW_MASK>(0.0, m.c), dpp_f64_keep<CTRL, ROW_MASK>(1.0, m.d)};
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ V2 dpp_v2_zero(const V2 &v) {
    return V2{dpp_f64_keep<CTRL, ROW_MASK>(0.0, v.x), dpp_f64_keep<CTRL, ROW_MASK>(0.0, v.y)};
}

__device__ __forceinline__ bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ------------------------------------------------------------------------------------------------
// Load / store T consecutive frames of one channel of a (frames, channels) float32 buffer.
template <int T>
__device__ __forceinline__ void load_frames(const float *base, int64_t f0, int64_t n, int channels, int ch,
                                            float (&x)[T]) {
    if (channels == 1 && f0 + T <= n && aligned16(base + f0)) {
#pragma unroll
        for (int j = 0; j < T; j += 4) {
            float4 t = *reinterpret_cast<const float4 *>(base + f0 + j);
            x[j] = t.x; x[j + 1] = t.y; x[j + 2] = t.z; x[j + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < T; ++j) x[j] = (f0 + j < n) ? base[(f0 + j) * channels + ch] : 0.0f;
    }
}

template <int T>
__device__ __forceinline__ void store_frames(float *base, int64_t f0, int64_t n, int channels, int ch,
                                             const float (&y)[T]) {
    if (channels == 1 && f0 + T <= n && aligned16(base + f0)) {
#pragma unroll
        for (int j = 0; j < T; j += 4)
            *reinterpret_cast<float4 *>(base + f0 + j) = make_float4(y[j], y[j + 1], y[j + 2], y[j + 3]);
    } else {
#pragma unroll
        for (int j = 0; j < T; ++j)
            if (f0 + j < n) base[(f0 + j) * channels + ch] = y[j];
    }
}

// Store T frames of a mono result replicated over `channels` output channels (np.tile).
template <int T>
__device__ __forceinline__ void store_frames_tiled(float *base, int64_t f0, int64_t n, int channels,
                                                   const float (&y)[T]) {
    if (channels == 1) {
        store_frames<T>(base, f0, n, 1, 0, y);
    } else {
#pragma unroll
        for (int j = 0; j < T; ++j)
            if (f0 + j < n)
                for (int c = 0; c < channels; ++c) base[(f0 + j) * channels + c] = y[j];
    }
}

// ------------------------------------------------------------------------------------------------
// Block-wide exclusive prefix sum of one double per thread (sequential-in-lane order), plus the
// block total.  `lds` must hold kWaves doubles.  Contains two __syncthreads.
__device__ __forceinline__ double block_excl_sum(double v, double *lds, double &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double o = __shfl_up(inc, d, 64);
        if (lane >= d) inc = o + inc;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    double woff = 0.0, tot = 0.0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        double t = lds[w];
        if (w < wave) woff = woff + t;
        tot = tot + t;
    }
    __syncthreads();
    total = tot;
    double ex = __shfl_up(inc, 1, 64);
    if (lane == 0) ex = 0.0;
    return woff + ex;
}

// Block-wide scan for the scalar constant map y' = lam^len * y + b.  On entry `e` is this thread's
// zero-state chunk response; lamp[k] = lam^(T*2^k) for k=0..5, lam_wave = lam^(T*64),
// lam_lane = lam^(T*lane).  Returns this thread's carry-in given the tile carry-in `carry`, and
// updates `carry` to the tile carry-out.  `lds` holds kWaves doubles.
__device__ __forceinline__ double block_scan_scalar_affine(double e, const double (&lamp)[6], double lam_wave,
                                                           double lam_lane, double *lds, double &carry) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double inc = e;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double o = __shfl_up(inc, 1 << k, 64);
        if (lane >= (1 << k)) inc = lamp[k] * o + inc;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    double cw = carry, cn = carry;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        double t = lds[w];
        if (w < wave) cw = lam_wave * cw + t;
        cn = lam_wave * cn + t;
    }
    __syncthreads();
    carry = cn;
    double ex = __shfl_up(inc, 1, 64);
    if (lane == 0) ex = 0.0;
    return lam_lane * cw + ex;
}

// Inclusive scans over a wave on DPP moves (VALU; a `__shfl_up` step is two ds_bpermute round trips through LDS):
// Kogge-Stone inside each 16-lane row (row_shr), then row 0 -> 1 and 2 -> 3 (lane 15 of the previous row), then
// rows 0-1 -> 2, 3 (lane 31).  Lanes without a source receive the neutral element.  Every oscillator kernel scans
// with these two, so their folds -- and bits -- agree with each other.
__device__ __forceinline__ double wave_incl_sum_dpp(double v) {
    v = dpp_f64_keep<0x111, 0xf>(0.0, v) + v;
    v = dpp_f64_keep<0x112, 0xf>(0.0, v) + v;
    v = dpp_f64_keep<0x114, 0xf>(0.0, v) + v;
    v = dpp_f64_keep<0x118, 0xf>(0.0, v) + v;
    v = dpp_f64_keep<0x142, 0xa>(0.0, v) + v;
    v = dpp_f64_keep<0x143, 0xc>(0.0, v) + v;
    return v;
}
// y' = lam^len * y + b: lamp[k] = lam^(T*2^k); lam16 / lam32 = lam^(T*((lane & 15) + 1)) / lam^(T*((lane & 31) + 1))
__device__ __forceinline__ double wave_incl_affine_dpp(double e, const double (&lamp)[6], double lam16, double lam32) {
    // (fused multiply-adds: these numbers only feed carries -- the samples themselves are re-run from the scanned
    // carry-in in the reference's operation order; k_blitsaw_chain folds with the same fused step)
    e = __builtin_fma(lamp[0], dpp_f64_keep<0x111, 0xf>(0.0, e), e);
    e = __builtin_fma(lamp[1], dpp_f64_keep<0x112, 0xf>(0.0, e), e);
    e = __builtin_fma(lamp[2], dpp_f64_keep<0x114, 0xf>(0.0, e), e);
    e = __builtin_fma(lamp[3], dpp_f64_keep<0x118, 0xf>(0.0, e), e);
    e = __builtin_fma(lam16, dpp_f64_keep<0x142, 0xa>(0.0, e), e);
    e = __builtin_fma(lam32, dpp_f64_keep<0x143, 0xc>(0.0, e), e);
    return e;
}
// the per-lane powers of a scan over chunks of T samples, from lamp[k] = lam^(T*2^k)
struct LanePowers {
    double lane, p16, p32;      // lam^(T*lane), lam^(T*((lane & 15) + 1)), lam^(T*((lane & 31) + 1))
};
__device__ __forceinline__ LanePowers lane_powers(const double (&lamp)[6], int lane) {
    LanePowers r{1.0, 1.0, 1.0};
    const int a = (lane & 15) + 1, b = (lane & 31) + 1;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (lane & (1 << k)) r.lane = r.lane * lamp[k];
        if (a & (1 << k)) r.p16 = r.p16 * lamp[k];
        if (b & (1 << k)) r.p32 = r.p32 * lamp[k];
    }
    return r;
}

// Wide forms for a workgroup of NW = 4*G waves that must reproduce, bit for bit, what a 4-wave workgroup
// computes on G consecutive tiles: the wave values are the same, so it is enough to fold them in the same
// order.  sum_carry: running sum entering the first of the G tiles (advanced to the one leaving the last).
template <int NW>
__device__ __forceinline__ double block_excl_sum_wide(double v, double *lds, double &sum_carry) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double inc = wave_incl_sum_dpp(v);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    double base = sum_carry, mine_base = sum_carry, woff = 0.0;
#pragma unroll
    for (int g = 0; g < NW / kWaves; ++g) {
        double tot = 0.0, w_local = 0.0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const double t = lds[g * kWaves + w];
            if (g * kWaves + w < wave) w_local = w_local + t;
            tot = tot + t;
        }
        if (g == wave / kWaves) {
            mine_base = base;
            woff = w_local;
        }
        base = base + tot;                                   // carry_sum = carry_sum + tile_total
    }
    __syncthreads();
    sum_carry = base;
    const double ex = dpp_f64_keep<0x138, 0xf>(0.0, inc);   // wave_shr:1 -- the lane before, 0 for lane 0
    return mine_base + (woff + ex);                          // chunk_base = carry_sum + off
}

template <int NW>
__device__ __forceinline__ double block_scan_scalar_affine_wide(double e, const double (&lamp)[6], double lam_wave,
                                                                double lam_lane, double *lds, double &carry) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double inc = e;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double o = __shfl_up(inc, 1 << k, 64);
        if (lane >= (1 << k)) inc = lamp[k] * o + inc;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    double cw = carry, cn = carry;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        double t = lds[w];
        if (w < wave) cw = lam_wave * cw + t;
        cn = lam_wave * cn + t;
    }
    __syncthreads();
    carry = cn;
    double ex = __shfl_up(inc, 1, 64);
    if (lane == 0) ex = 0.0;
    return lam_lane * cw + ex;
}

// block_excl_sum_wide when every thread contributes the SAME value (a scalar-frequency oscillator on a tile
// whose frames are all live): every wave's total is the same number, so the folds need no exchange and no
// barrier.  Same operations in the same order as the exchanging form, hence the same bits.
// The two numbers block_excl_sum_wide_uniform derives from v alone: this thread's offset inside the tile and what the
// carry advances by per group of kWaves waves.  With them a tile's prefix is `carry + offset` (groups before the
// thread's own first advance the carry): a caller whose v never changes makes them once -- same operations, same bits.
struct UniformPrefix {
    double offset, tot;
};
__device__ __forceinline__ UniformPrefix uniform_prefix(double v) {
    const int wave = threadIdx.x >> 6;
    const double inc = wave_incl_sum_dpp(v);
    const double t = readlane_f64(inc, 63);
    double tot = 0.0, w_local = 0.0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        if (w < (wave & (kWaves - 1))) w_local = w_local + t;
        tot = tot + t;
    }
    const double ex = dpp_f64_keep<0x138, 0xf>(0.0, inc);
    return UniformPrefix{w_local + ex, tot};
}
template <int NW>
__device__ __forceinline__ double block_excl_sum_wide_uniform(const UniformPrefix &u, double &sum_carry) {
    const int wave = threadIdx.x >> 6;
    double base = sum_carry, mine_base = sum_carry;
#pragma unroll
    for (int g = 0; g < NW / kWaves; ++g) {
        if (g == wave / kWaves) mine_base = base;
        base = base + u.tot;
    }
    sum_carry = base;
    return mine_base + u.offset;
}
template <int NW>
__device__ __forceinline__ double block_excl_sum_wide_uniform(double v, double &sum_carry) {
    return block_excl_sum_wide_uniform<NW>(uniform_prefix(v), sum_carry);
}

// block_scan_scalar_affine_wide with ONE barrier: `lds` holds two images of NW doubles used alternately
// (`parity` flips on every call), so the readers of one image are separated from its next writer by the
// barrier of the call in between.
template <int NW>
__device__ __forceinline__ double block_scan_scalar_affine_wide1(double e, const double (&lamp)[6], double lam_wave,
                                                                 const LanePowers &lp, double *lds, int parity,
                                                                 double &carry) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *img = lds + (parity & 1) * NW;
    const double inc = wave_incl_affine_dpp(e, lamp, lp.p16, lp.p32);
    if (lane == 63) img[wave] = inc;
    __syncthreads();
    double cw = carry, cn = carry;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        double t = img[w];
        if (w < wave) cw = __builtin_fma(lam_wave, cw, t);
        cn = __builtin_fma(lam_wave, cn, t);
    }
    carry = cn;
    const double ex = dpp_f64_keep<0x138, 0xf>(0.0, inc);
    return __builtin_fma(lp.lane, cw, ex);
}

// ================================================================================================
// BiquadPE, constant coefficients
// ================================================================================================
constexpr int kBqT = 16;                       // frames per thread per tile
constexpr int kBqTile = kBlock * kBqT;         // 4096 frames
constexpr int kBqMaxSeg = 1024;
constexpr int kBqPow = 11;                     // binary powers of A^segment kept for the fold

struct BqShared {
    M2 pstep[6];       // A^(T*2^k)
    M2 pwave;          // A^(T*64)
    M2 ptile;          // A^(T*256)
    M2 pseg[kBqPow];   // (A^segment)^(2^k)
    V2 wave_tot[kWaves];
    V2 red[kWaves];
};

// MODE 0: reduce (write the segment's zero-state response to agg); MODE 1: apply (produce output).
template <int MODE>
__global__ void __launch_bounds__(kBlock)
k_biquad_const(float *out, int64_t out_stride, const float *in, int64_t in_stride, int64_t n, int channels,
               const double *coef, double *state, const double *state_snapshot, double *agg, int seg_tiles,
               int nseg) {
    __shared__ BqShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chain = blockIdx.y;
    const int inst = chain / channels, ch = chain - inst * channels;
    const int seg = blockIdx.x;

    const double b0 = coef[inst * 5 + 0], b1 = coef[inst * 5 + 1], b2 = coef[inst * 5 + 2];
    const double a1 = coef[inst * 5 + 3], a2 = coef[inst * 5 + 4];

    // ---- loop-invariant matrix powers (thread 0 -> LDS) ----
    if (tid == 0) {
        M2 A{-a1, 1.0, -a2, 0.0};
        M2 p = A;
#pragma unroll
        for (int s = 1; s < kBqT; s <<= 1) p = mm(p, p);     // A^T (T is a power of two)
        for (int k = 0; k < 6; ++k) {
            sh.pstep[k] = p;
            p = mm(p, p);
        }
        sh.pwave = p;                                        // A^(64 T)
        p = mm(p, p);
        p = mm(p, p);
        sh.ptile = p;                                        // A^(256 T)
        if (nseg > 1) {
            // A^segment = ptile^seg_tiles (binary exponentiation), then its binary powers.
            M2 r = m_identity(), q = p;
            for (int e = seg_tiles; e > 0; e >>= 1) {
                if (e & 1) r = mm(r, q);
                q = mm(q, q);
            }
            for (int k = 0; k < kBqPow; ++k) {
                sh.pseg[k] = r;
                r = mm(r, r);
            }
        }
    }
    __syncthreads();

    M2 mlane = m_identity();                                  // A^(T*lane)
#pragma unroll
    for (int k = 0; k < 6; ++k)
        if (lane & (1 << k)) mlane = mm(sh.pstep[k], mlane);

    // ---- segment carry-in ----
    V2 carry{0.0, 0.0};
    if (MODE == 1) {
        if (nseg == 1) {
            carry = V2{state[chain * 2 + 0], state[chain * 2 + 1]};
        } else {
            // carry = (A^seg)^seg_index * z_init + sum_{j<seg} (A^seg)^(seg-1-j) * agg_j
            V2 part{0.0, 0.0};
            for (int j = tid; j <= seg; j += kBlock) {
                V2 v;
                int e;
                if (j == seg) {                               // the initial-state term
                    v = V2{state_snapshot[chain * 2 + 0], state_snapshot[chain * 2 + 1]};
                    e = seg;
                } else {
                    v = V2{agg[((int64_t)chain * nseg + j) * 2 + 0], agg[((int64_t)chain * nseg + j) * 2 + 1]};
                    e = seg - 1 - j;
                }
                for (int k = 0; k < kBqPow; ++k)
                    if (e & (1 << k)) v = mv(sh.pseg[k], v);
                part = vadd(part, v);
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                part.x += __shfl_down(part.x, d, 64);
                part.y += __shfl_down(part.y, d, 64);
            }
            if (lane == 0) sh.red[wave] = part;
            __syncthreads();
            carry = V2{0.0, 0.0};
#pragma unroll
            for (int w = 0; w < kWaves; ++w) carry = vadd(carry, sh.red[w]);
            __syncthreads();
        }
    }

    const float *ib = in + (int64_t)inst * in_stride;
    float *ob = out + (int64_t)inst * out_stride;
    V2 final_state{0.0, 0.0};
    bool have_final = false;

    const int64_t tile0 = (int64_t)seg * seg_tiles;
    for (int t = 0; t < seg_tiles; ++t) {
        const int64_t base = (tile0 + t) * kBqTile;
        if (base >= n) break;
        const int64_t f0 = base + (int64_t)tid * kBqT;
        float xf[kBqT];
        load_frames<kBqT>(ib, f0, n, channels, ch, xf);

        // pass 1: zero-state response of this chunk
        V2 e{0.0, 0.0};
#pragma unroll
        for (int j = 0; j < kBqT; ++j) {
            double x = (double)xf[j];
            double y = e.x + b0 * x;
            double z0 = (e.y + b1 * x) - a1 * y;
            e.y = b2 * x - a2 * y;
            e.x = z0;
        }
        // wave scan (offset vectors only; the matrices are the LDS-resident powers)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            V2 o = shfl_up_v2(e, 1 << k);
            if (lane >= (1 << k)) e = vadd(mv(sh.pstep[k], o), e);
        }
        if (lane == 63) sh.wave_tot[wave] = e;
        __syncthreads();
        V2 cw = carry, cn = carry;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            V2 tot = sh.wave_tot[w];
            if (w < wave) cw = vadd(mv(sh.pwave, cw), tot);
            cn = vadd(mv(sh.pwave, cn), tot);
        }
        __syncthreads();
        carry = cn;

        if (MODE == 1) {
            V2 ex = shfl_up_v2(e, 1);
            if (lane == 0) ex = V2{0.0, 0.0};
            V2 z = vadd(mv(mlane, cw), ex);
            float yf[kBqT];
            // pass 2: scipy lfilter DF-II-T operation order from the scanned carry-in
#pragma unroll
            for (int j = 0; j < kBqT; ++j) {
                double x = (double)xf[j];
                double y = z.x + b0 * x;
                double z0 = (z.y + b1 * x) - a1 * y;
                z.y = b2 * x - a2 * y;
                z.x = z0;
                yf[j] = (float)y;
                if (f0 + j == n - 1) {
                    final_state = z;
                    have_final = true;
                }
            }
            store_frames<kBqT>(ob, f0, n, channels, ch, yf);
        }
    }
    if (MODE == 0) {
        if (tid == 0) {
            agg[((int64_t)chain * nseg + seg) * 2 + 0] = carry.x;
            agg[((int64_t)chain * nseg + seg) * 2 + 1] = carry.y;
            if (seg == 0) {
                // snapshot of the carried state for the apply launch (whose last segment overwrites
                // `state` while other segments may not have started yet)
                double *snap = const_cast<double *>(state_snapshot);
                snap[chain * 2 + 0] = state[chain * 2 + 0];
                snap[chain * 2 + 1] = state[chain * 2 + 1];
            }
        }
    } else if (have_final) {
        state[chain * 2 + 0] = final_state.x;
        state[chain * 2 + 1] = final_state.y;
    }
}

// ------------------------------------------------------------------------------------------------
// Settled single launch.  A stable section forgets: once every entry of A^W is below 2^-90 the state W
// frames back no longer reaches the float64 arithmetic of the present.  Each workgroup then recovers
// its carry-in by running the recurrence from zero over the frames that precede its range (outputs
// discarded) instead of waiting for its predecessors: one launch, no cross-workgroup traffic, every
// input frame fetched from HBM once (the warm-up frames are L2/MALL hits of a neighbour's fetch).
// W is evaluated by the host from the coefficients (pgx_biquad_const's settle_frames); slowly decaying
// sections (W above kSbMaxWarm half-tiles) keep the exact reduce + apply pair.
//
// Geometry: 512 threads x 16 frames = one 8192-frame tile, addressed in 4096-frame halves so that the
// smallest plan (one half of warm-up + one half of output per workgroup) is a single scan step.
// Workgroup 0 renders the halves [0, head) and the tail [tail_start, halves): it alone reads the
// carried state (first thing) and writes it (last thing); workgroup g >= 1 renders seg halves from
// head + (g-1)*seg.  head >= warm, so no other workgroup ever needs the carried state.
// ------------------------------------------------------------------------------------------------
constexpr int kSbBlock = 512;
constexpr int kSbWaves = kSbBlock / 64;
constexpr int kSbHalf = (kSbBlock / 2) * kBqT;     // 4096 frames
constexpr int kSbMaxWarm = 16;                     // half-tiles

// DPP moves of a double (two 32-bit halves).  CTRL: 0x110+n = row_shr:n (within 16-lane rows),
// 0x138 = wave_shr:1, 0x142 = row_bcast:15, 0x143 = row_bcast:31.  Lanes without a source, or outside
// ROW_MASK, receive 0.0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ V2 dpp_v2(const V2 &v) {
    return V2{dpp_f64<CTRL, ROW_MASK>(v.x), dpp_f64<CTRL, ROW_MASK>(v.y)};
}

struct SbShared {
    V2 wave_tot[2][kSbWaves];        // two images used in turn: one barrier per tile (the readers of one image are
};                                   // separated from its next writer by the barrier of the tile in between)

// Staging of a wave's 1024 frames through wave-private LDS: HBM is touched with fully coalesced 16-byte
// accesses (lane l of access i owns frames i*256 + 4l ..), the scan wants 16 consecutive frames per lane.
// Chunk c (16 frames) lives at word c*20: the 4-word pad makes both access patterns bank-conflict free.
constexpr int kStageWords = 64 * 20;

__device__ __forceinline__ void stage_fetch(const float *src, int lane, float (&t)[kBqT]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v = *reinterpret_cast<const float4 *>(src + i * 256 + lane * 4);
        t[4 * i] = v.x; t[4 * i + 1] = v.y; t[4 * i + 2] = v.z; t[4 * i + 3] = v.w;
    }
}
__device__ __forceinline__ int stage_slot(int i, int lane) {       // LDS word of frame i*256 + 4*lane
    const int idx = i * 256 + lane * 4;
    return (idx >> 4) * 20 + (idx & 15);
}
// coalesced order (as fetched) -> 16 consecutive frames per lane
__device__ __forceinline__ void stage_to_chunks(float *lds, int lane, float (&t)[kBqT]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4 *>(lds + stage_slot(i, lane)) = make_float4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v = *reinterpret_cast<const float4 *>(lds + lane * 20 + 4 * i);
        t[4 * i] = v.x; t[4 * i + 1] = v.y; t[4 * i + 2] = v.z; t[4 * i + 3] = v.w;
    }
}
// 16 consecutive frames per lane -> coalesced stores
__device__ __forceinline__ void stage_store(float *lds, float *dst, int lane, const float (&y)[kBqT]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4 *>(lds + lane * 20 + 4 * i) = make_float4(y[4 * i], y[4 * i + 1], y[4 * i + 2], y[4 * i + 3]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4 *>(dst + i * 256 + lane * 4) = *reinterpret_cast<const float4 *>(lds + stage_slot(i, lane));
}

// Matrix-power tables of one section, computed once per coefficient set (pgx_biquad_tables):
// A^(16*2^k) for k = 0..5, A^(16*64), then A^(16*j) for j = 0..63.
constexpr int kBqTableDoubles = 28 + 4 * 64 + 2 * kBqT;     // ... then the first rows of A^j, j = 0..kBqT-1 (SINE pass 2)
constexpr int kBqRowsAt = 28 + 4 * 64;

__global__ void __launch_bounds__(64)
k_biquad_tables(double *tables, const double *coef) {
    __shared__ M2 ps[6];
    const int lane = threadIdx.x, inst = blockIdx.x;
    double *tb = tables + (int64_t)inst * kBqTableDoubles;
    if (lane == 0) {
        M2 p{-coef[inst * 5 + 3], 1.0, -coef[inst * 5 + 4], 0.0};
#pragma unroll
        for (int s = 1; s < kBqT; s <<= 1) p = mm(p, p);
        for (int k = 0; k < 6; ++k) {
            ps[k] = p;
            tb[4 * k] = p.a; tb[4 * k + 1] = p.b; tb[4 * k + 2] = p.c; tb[4 * k + 3] = p.d;
            p = mm(p, p);
        }
        tb[24] = p.a; tb[25] = p.b; tb[26] = p.c; tb[27] = p.d;
    }
    __syncthreads();
    M2 m = m_identity();
#pragma unroll
    for (int k = 0; k < 6; ++k)
        if (lane & (1 << k)) m = mm(ps[k], m);
    double *ml = tb + 28 + 4 * lane;
    ml[0] = m.a; ml[1] = m.b; ml[2] = m.c; ml[3] = m.d;
    if (lane == 0) {
        // first rows of A^j: what a carried state z contributes to the output of the j-th frame after it, y_h[j] =
        // (A^j z).x (the sine-source variant adds it to the zero-state outputs it kept from pass 1)
        M2 q = m_identity();
        const M2 a1m{-coef[inst * 5 + 3], 1.0, -coef[inst * 5 + 4], 0.0};
        for (int j = 0; j < kBqT; ++j) {
            tb[kBqRowsAt + 2 * j] = q.a;
            tb[kBqRowsAt + 2 * j + 1] = q.b;
            q = mm(a1m, q);
        }
    }
}

__device__ __forceinline__ M2 load_m2(const double *p) { return M2{p[0], p[1], p[2], p[3]}; }

// p*v + q with fused multiply-adds.  Used only where the result feeds a carry (zero-state responses and
// their scan), never in the output pass, whose operation order is the reference's.
__device__ __forceinline__ V2 mv_add_fma(const M2 &p, const V2 &v, const V2 &q) {
    return V2{__builtin_fma(p.a, v.x, __builtin_fma(p.b, v.y, q.x)),
              __builtin_fma(p.c, v.x, __builtin_fma(p.d, v.y, q.y))};
}

// SINE: the input is a pure SinePE (sine_pe.py:159-175) generated in registers instead of read from HBM: the chain
// BiquadPE(SinePE) then moves 4 B per frame (its output) in one launch.  A thread's first frame is evaluated exactly
// as k_sine evaluates it (phase0 + w * (n / sr), pgx_sincos_bounded); its other 15 frames turn that (sin, cos) pair
// by the fixed angle w / sr (two multiplies and two fused multiply-adds per frame, error ~1e-16 per step).  The
// reference's own phase carries ~ulp(phase) of rounding noise per sample; the rotated samples sit inside it.
struct SbSine {
    double w, amp, phase0, sr, inv_sr, cos_d, sin_d;
    double two_cos_d;          // 2 cos(w / sr) for the three-term recurrence, or 0: rotate (very low frequencies)
    double tile_cos, tile_sin; // cos / sin of a tile's advance, BLOCK * 16 frames of w / sr (the angle reduced on the host)
    int64_t start;
    double *state_backup;      // receives the carried state on entry (a look-ahead window's snapshot), or nullptr
};

#ifndef PGX_SB_SINE_WAVES
#define PGX_SB_SINE_WAVES 3        // waves per SIMD the sine-source variant is compiled for: 166 VGPRs, no spill (4: 128 + 28 B of scratch per lane, 6 MB of extra HBM traffic per 33 M frames and 5 % slower)
#endif
// BLOCK: threads per workgroup -- 512 (a tile = two halves), or 256 for the sine-source variant (a tile = one half: at its
// 166 VGPRs a CU holds one 8-wave workgroup, whose single barrier per tile then idles the whole CU, or three 4-wave ones)
template <bool MONO, bool STAGED, bool SINE = false, int BLOCK = kSbBlock>
__global__ void __launch_bounds__(BLOCK, MONO ? (SINE ? PGX_SB_SINE_WAVES : 4) : 2)
k_biquad_settled(float *__restrict__ out, int64_t out_stride, const float *__restrict__ in, int64_t in_stride,
                 int64_t n, int channels_arg, const double *__restrict__ coef, const double *__restrict__ tables,
                 double *state, int seg, int head, int tail, int warm, int groups, SbSine sine = SbSine{}) {
    constexpr int kTileHalves = BLOCK * kBqT / kSbHalf;         // halves a tile covers: 2, or 1
    __shared__ SbShared sh;
    __shared__ __attribute__((aligned(16))) float stage_lds[STAGED ? (BLOCK / 64) * kStageWords : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int channels = MONO ? 1 : channels_arg;
    const int chain = blockIdx.y;
    const int inst = chain / channels, ch = MONO ? 0 : chain - inst * channels;
    // Workgroups are dealt to the 8 XCDs round-robin and each XCD has its own L2.  While the warm-up is
    // a sizeable part of a segment, give every XCD one contiguous run of segments, so that a segment's
    // warm-up frames are its left neighbour's L2 lines (measured: 1M frames fetch 4.3 MB instead of 8.2 MB).
    // Long segments keep the round-robin order (the re-read is ~3% there and it streams ~10% faster).
    const int per_xcd = gridDim.x >> 3;                        // gridDim.x is a multiple of 8
    const int g = seg <= 8 ? (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    if (g >= groups) return;
    const float *ib = in + (int64_t)inst * in_stride;
    float *ob = out + (int64_t)inst * out_stride;
    float *wlds = stage_lds + (STAGED ? wave * kStageWords : 0);
    // STAGED (mono only): a wave whose 1024 frames are all inside the block and 16-byte aligned moves them
    // with coalesced accesses; any other wave takes the per-lane path.  `xn_staged` says which layout xn has.
    const bool io_aligned = STAGED && (SINE || aligned16(ib)) && aligned16(ob);
    bool xn_staged = false;
    const int64_t halves = (n + kSbHalf - 1) / kSbHalf;
    int64_t tail_start = halves - tail;
    if (tail_start < head) tail_start = head;

    // the ranges of this workgroup, in halves: [hb0, he0) and, for workgroup 0 only, [hb1, he1)
    int64_t hb = g == 0 ? 0 : head + (int64_t)(g - 1) * seg;
    int64_t he = g == 0 ? (head < halves ? head : halves) : (hb + seg < tail_start ? hb + seg : tail_start);
    int64_t h = hb == 0 ? 0 : hb - warm;
    int64_t emit_to = he * kSbHalf < n ? he * kSbHalf : n;

    // first tile's frames are requested before anything else so that their latency covers the table loads
    float xn[kBqT];
    int64_t anchor_tile = -(int64_t)1 << 40;                   // SINE: the tile the anchor (sin, cos) below belongs to
    double anchor_s = 0.0, anchor_c = 1.0;
    auto request = [&](int64_t hh) {
        const int64_t w0 = hh * kSbHalf + (int64_t)(tid - lane) * kBqT;      // first frame of this wave
        const int64_t f0 = w0 + lane * kBqT;
        if (SINE) {
            xn_staged = false;
            // The thread's first frame: k_sine's evaluation for the first tile of a run of consecutive tiles, that pair
            // turned by a tile's advance for every further one (4 operations instead of the division and the sincos, ~35;
            // a workgroup's run is a dozen tiles: ~1e-15).
            double sn, cs;
            if (PGX_HOT(hh == anchor_tile + kTileHalves)) {
                sn = __builtin_fma(anchor_s, sine.tile_cos, anchor_c * sine.tile_sin);
                cs = __builtin_fma(anchor_c, sine.tile_cos, -(anchor_s * sine.tile_sin));
            } else {
                const double t = pgx::pgx_div_by((double)(sine.start + f0), sine.sr, sine.inv_sr);
                pgx::pgx_sincos_bounded(sine.phase0 + sine.w * t, sn, cs);
            }
            anchor_tile = hh;
            anchor_s = sn;
            anchor_c = cs;
            if (PGX_HOT(sine.two_cos_d != 0.0)) {
                // three-term recurrence sin(p + (j+1)d) = 2 cos(d) sin(p + jd) - sin(p + (j-1)d): ONE fused multiply-add
                // per frame instead of the four operations of the rotation (the cosine is not needed).  Its error
                // after k steps is <= k ulp / d: 15 steps, d >= 1e-3 (the host's condition) -> below 2e-12
                double s0 = sn, s1 = __builtin_fma(cs, sine.sin_d, sn * sine.cos_d);
#ifndef PGX_C2_NO_UNIT_AMP
                if (PGX_HOT(sine.amp == 1.0)) {
                    // SinePE's default amplitude (C2's): amp * s is s, bit for bit -- sixteen multiplications a tile less
                    // (a wave-uniform choice: sine.amp is a kernel argument)
                    xn[0] = (float)s0;
                    xn[1] = (float)s1;
#pragma unroll
                    for (int j = 2; j < kBqT; ++j) {
                        const double s2 = __builtin_fma(sine.two_cos_d, s1, -s0);
                        xn[j] = (float)s2;
                        s0 = s1;
                        s1 = s2;
                    }
                    return;
                }
#endif
                xn[0] = (float)(sine.amp * s0);
                xn[1] = (float)(sine.amp * s1);
#pragma unroll
                for (int j = 2; j < kBqT; ++j) {
                    const double s2 = __builtin_fma(sine.two_cos_d, s1, -s0);
                    xn[j] = (float)(sine.amp * s2);
                    s0 = s1;
                    s1 = s2;
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < kBqT; ++j) {
                xn[j] = (float)(sine.amp * sn);
                const double s2 = __builtin_fma(cs, sine.sin_d, sn * sine.cos_d);
                cs = __builtin_fma(-sn, sine.sin_d, cs * sine.cos_d);
                sn = s2;
            }
            return;
        }
        xn_staged = STAGED && PGX_HOT(io_aligned && w0 + 64 * kBqT <= emit_to);
        if (xn_staged) {
            stage_fetch(ib + w0, lane, xn);
        } else if (f0 < emit_to) {
            load_frames<kBqT>(ib, f0, n, channels, ch, xn);
        } else {
#pragma unroll
            for (int j = 0; j < kBqT; ++j) xn[j] = 0.0f;
        }
    };
    if (!SINE && hb < he) request(h);

    const double b0 = coef[inst * 5 + 0], b1 = coef[inst * 5 + 1], b2 = coef[inst * 5 + 2];
    const double a1 = coef[inst * 5 + 3], a2 = coef[inst * 5 + 4];
    // uniform loads -> scalar registers
    const double *tb = tables + (int64_t)inst * kBqTableDoubles;
    M2 pstep[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pstep[k] = load_m2(tb + 4 * k);
    const M2 pwave = load_m2(tb + 24);
    // per-lane powers: carry-in of the wave -> lane, previous row(s) -> lane
    const M2 mlane = load_m2(tb + 28 + 4 * lane);
    const M2 m16 = load_m2(tb + 28 + 4 * ((lane & 15) + 1));
    const M2 m32 = load_m2(tb + 28 + 4 * ((lane & 31) + 1));
    // SINE: the first rows of A^j (pass 2) live in LDS, read as broadcasts where they are used.  As 32 uniform values in
    // scalar registers they did not fit beside the scan's matrices: the compiler parked them in the lanes of a vector
    // register and fetched them back with 64 v_readlane per tile -- a sixth of the loop's vector instructions (end of round 4)
    __shared__ __attribute__((aligned(16))) double rows_lds[SINE ? 2 * kBqT : 2];
    if (SINE) {
        if (tid < 2 * kBqT) rows_lds[tid] = tb[kBqRowsAt + tid];
        __syncthreads();
    }

    int image = 0;
#pragma nounroll
    for (int range = 0; range < 2; ++range) {
        if (range == 1) {
            if (g != 0) break;
            hb = tail_start;
            he = halves;
            if (hb >= he) break;
            h = hb - warm;
            emit_to = n;
            if (!SINE) request(h);
        }
        if (hb >= he) continue;
        V2 carry{0.0, 0.0};
        if (hb == 0) {
            carry = V2{state[chain * 2 + 0], state[chain * 2 + 1]};
            if (SINE && sine.state_backup && tid == 0) {          // workgroup 0 alone reads and writes the state
                sine.state_backup[chain * 2 + 0] = carry.x;
                sine.state_backup[chain * 2 + 1] = carry.y;
            }
        }
        const int64_t emit_from = hb * kSbHalf;

#pragma nounroll
        for (; h < he; h += kTileHalves) {
            const int64_t f0 = h * kSbHalf + (int64_t)tid * kBqT;
            float xf[kBqT];
            if (SINE) {
                request(h);                                    // generated, not fetched: nothing to have in flight
#pragma unroll
                for (int j = 0; j < kBqT; ++j) xf[j] = xn[j];
            } else {
#pragma unroll
                for (int j = 0; j < kBqT; ++j) xf[j] = xn[j];
                if (STAGED && xn_staged) stage_to_chunks(wlds, lane, xf);
                if (h + kTileHalves < he) request(h + kTileHalves);                // next tile's frames, in flight during this one
            }
            // zero-state response of the 16-frame chunk
            V2 e{0.0, 0.0};
            const double na1 = -a1, na2 = -a2;
            double yz[SINE ? kBqT : 1];                        // SINE: the zero-state outputs stay for pass 2
#pragma unroll
            for (int j = 0; j < kBqT; ++j) {
                const double x = (double)xf[j];
                const double y = __builtin_fma(b0, x, e.x);
                e.x = __builtin_fma(na1, y, __builtin_fma(b1, x, e.y));
                e.y = __builtin_fma(na2, y, b2 * x);
                if (SINE) yz[j] = y;
            }
            if (!SINE) {
#pragma unroll
                for (int j = 0; j < kBqT; ++j) asm volatile("" : "+v"(xf[j]));   // re-convert in pass 2, keep VGPRs low
            }
            // inclusive scan over the wave: Kogge-Stone inside each 16-lane row (DPP row shifts) ...
            e = mv_add_fma(pstep[0], dpp_v2<0x111, 0xf>(e), e);
            e = mv_add_fma(pstep[1], dpp_v2<0x112, 0xf>(e), e);
            e = mv_add_fma(pstep[2], dpp_v2<0x114, 0xf>(e), e);
            e = mv_add_fma(pstep[3], dpp_v2<0x118, 0xf>(e), e);
            // ... then row 0 -> 1, row 2 -> 3 (lane 15 of the previous row), then rows 0-1 -> 2, 3 (lane 31)
            e = mv_add_fma(m16, dpp_v2<0x142, 0xa>(e), e);
            e = mv_add_fma(m32, dpp_v2<0x143, 0xc>(e), e);
            V2 *tot = sh.wave_tot[image];
            image ^= 1;
            if (lane == 63) tot[wave] = e;
            __syncthreads();
            V2 cw = carry, run = carry;
#pragma unroll
            for (int w = 0; w < BLOCK / 64; ++w) {
                run = mv_add_fma(pwave, run, tot[w]);
                if (w + 1 == wave) cw = run;                   // wave-uniform pick of this wave's carry-in
            }
#ifdef PGX_SB_TWO_BARRIERS
            __syncthreads();
#endif
            carry = run;

            if (PGX_HOT(f0 >= emit_from && f0 < emit_to)) {
                const V2 ex = dpp_v2<0x138, 0xf>(e);           // previous lane's inclusive value, 0 for lane 0
                const V2 zin = mv_add_fma(mlane, cw, ex);
                V2 z = zin;
                float yf[kBqT];
                if (SINE) {
                    // the zero-state outputs of pass 1 plus what the carried state adds, (A^j zin).x: two
                    // multiply-adds per frame instead of the recurrence again (the sine chain is within the rounding
                    // noise of the reference's phase anyway; the plain filter below keeps scipy's operation order)
                    const double *rows = rows_lds;               // (LDS broadcasts: see above)
#pragma unroll
                    for (int j = 0; j < kBqT; ++j)
                        yf[j] = (float)__builtin_fma(rows[2 * j], zin.x, __builtin_fma(rows[2 * j + 1], zin.y, yz[j]));
                } else {
                // scipy lfilter DF-II-T operation order from the scanned carry-in
#pragma unroll
                for (int j = 0; j < kBqT; ++j) {
                    double x = (double)xf[j];
                    double y = z.x + b0 * x;
                    double z0 = (z.y + b1 * x) - a1 * y;
                    z.y = b2 * x - a2 * y;
                    z.x = z0;
                    yf[j] = (float)y;
                }
                }
                const int64_t w0 = f0 - lane * kBqT;
                if (STAGED && PGX_HOT(io_aligned && w0 >= emit_from && w0 + 64 * kBqT <= emit_to))
                    stage_store(wlds, ob + w0, lane, yf);
                else
                    store_frames<kBqT>(ob, f0, n, channels, ch, yf);
                if (PGX_COLD(n - 1 - f0 < kBqT)) {               // the chunk holding the last frame: new state
                    z = zin;
                    for (int j = 0; j <= (int)(n - 1 - f0); ++j) {
                        double x = (double)xf[j];
                        double y = z.x + b0 * x;
                        double z0 = (z.y + b1 * x) - a1 * y;
                        z.y = b2 * x - a2 * y;
                        z.x = z0;
                    }
                    state[chain * 2 + 0] = z.x;
                    state[chain * 2 + 1] = z.y;
                }
            }
        }
    }
}

struct BqPlan {
    int seg_tiles;
    int nseg;
};

struct BqSettledPlan {
    bool ok;
    int seg, head, tail, warm, groups;        // in 4096-frame halves
};

BqSettledPlan biquad_settled_plan(int batch, int64_t n, int channels, int64_t settle_frames, bool have_tables,
                                  int workgroups = 512) {
    BqSettledPlan p{};
    if (!have_tables) return p;
    const int64_t halves = pgx::ceil_div(n, kSbHalf);
    const int64_t chains = (int64_t)batch * channels;
    const int64_t warm = pgx::ceil_div(settle_frames, kSbHalf);
    if (settle_frames > 0 && warm <= kSbMaxWarm) {
        int64_t want = workgroups / chains;                    // 512: two resident workgroups per CU, one round
        if (want < 1) want = 1;
        int64_t seg = pgx::ceil_div(halves, want);
        if (seg < warm) seg = warm;                            // warm-up never exceeds the rendered part
        const int64_t head = seg / 2 > warm ? seg / 2 : warm;  // workgroup 0: head + tail ~ one segment
        const int64_t tail = seg - head > 1 ? seg - head : 1;
        if (halves > head + tail) {
            p.ok = true;
            p.seg = (int)seg;
            p.head = (int)head;
            p.tail = (int)tail;
            p.warm = (int)warm;
            p.groups = 1 + (int)pgx::ceil_div(halves - head - tail, seg);
            return p;
        }
    }
    // One workgroup per chain over the whole block: nothing is assumed (the carried state is read, no
    // warm-up), this is just the faster tile engine.  Worth it when the chains alone fill the machine
    // (voice banks) or the block is a tile or two; long lone chains keep the segment-parallel exact pair.
    if (chains >= 64 || halves <= 2) {
        p.ok = true;
        p.seg = (int)halves;
        p.head = (int)halves;
        p.tail = 0;
        p.warm = 0;
        p.groups = 1;
    }
    return p;
}

BqPlan biquad_plan(int batch, int64_t n, int channels) {
    int64_t tiles = pgx::ceil_div(n, kBqTile);
    int64_t chains = (int64_t)batch * channels;
    int64_t want = 512 / chains;                  // aim for ~512 workgroups (2 per CU) in flight
    if (want < 1) want = 1;
    if (want > kBqMaxSeg) want = kBqMaxSeg;
    if (want > tiles) want = tiles;
    BqPlan p;
    p.seg_tiles = (int)pgx::ceil_div(tiles, want);
    p.nseg = (int)pgx::ceil_div(tiles, p.seg_tiles);
    return p;
}

// ================================================================================================
// BlitSawPE
// ================================================================================================
constexpr int kSawT = 8;
constexpr int kSawTile = kBlock * kSawT;   // 2048 frames
constexpr double kPi = 3.141592653589793;

constexpr int kSawWideWaves = 8;                  // 512-thread form for a handful of oscillators (203 VGPRs: 2 waves per SIMD)
struct SawShared {
    double sum[kSawWideWaves];
    double aff[2 * kSawWideWaves];        // two images: block_scan_scalar_affine_wide1
};

// Per-sample constants of the Dirichlet kernel derived from the frequency (blit_saw_pe.py:166-173,196).
struct SawConst {
    double inc, m, P, invP;
};
__device__ __forceinline__ SawConst saw_const(double f, double sr, double m_param, bool has_m, double m_stream) {
    SawConst c;
    c.inc = f / sr;
    const double fmax1 = f > 1.0 ? f : 1.0;                    // np.maximum(freq, 1.0)
    if (has_m) {
        int mi = (int)m_stream;                                // astype(int32): truncation
        c.m = (double)(mi > 1 ? mi : 1);
    } else if (m_param > 0.0) {
        int mi = (int)m_param;
        c.m = (double)(mi > 1 ? mi : 1);
    } else {
        double m_float = sr / (2.0 * fmax1);
        int mi = (int)floor(m_float);
        mi = mi - (1 - (mi % 2));
        c.m = (double)(mi > 1 ? mi : 1);
    }
    c.P = sr / fmax1;
    c.invP = 1.0 / c.P;
    return c;
}

// sin(m*theta) / sin(theta) of the Dirichlet kernel.  theta = pi * phase is below pi; m*theta stays below 3e6 for
// every M the rule sr / (2f) can produce (and for explicit M up to 9e5): then the bounded sine applies and the
// samples of a thread interleave.  BOUNDED is a workgroup-uniform choice made outside the sample loops.
template <bool BOUNDED>
__device__ __forceinline__ double saw_sin_num(double m_theta) {
    return BOUNDED ? pgx::pgx_sin_bounded(m_theta) : pgx::pgx_sin(m_theta);
}
constexpr double kSawBoundedM = 9.0e5;

// Scalar frequency, odd M (what the rule sr / (2f) always produces): a thread's 8 consecutive samples advance
// the phase by the same increment, so only the first one evaluates sin / cos of theta and M*theta; the other seven
// turn both pairs by the fixed angles pi*inc and M*pi*inc (two FMAs per component).  The wrap of the phase at 1
// needs no attention: theta -> theta - pi flips the sign of sin(theta) and, M being odd, of sin(M*theta): the
// ratio is unchanged.  71 -> ~40 instructions per sample in the Dirichlet block; the seven rotations add ~1e-15.
struct SawRot {
    double sd, cd, sm, cm;
    bool usable;
};
__device__ __forceinline__ SawRot saw_rot(const SawConst &k) {
    SawRot r;
    const double d = kPi * k.inc;
    pgx::pgx_sincos_bounded(d, r.sd, r.cd);
    pgx::pgx_sincos_bounded(k.m * d, r.sm, r.cm);
    const double half = k.m * 0.5;
    r.usable = k.m < kSawBoundedM && half != floor(half) && k.inc >= 0.0 && k.inc <= 0.5;
    return r;
}
// xb[j] = blit - 1/P for the thread's 8 samples; returns nothing else: the carried phase is taken elsewhere.
// INNER: the caller knows that all of the thread's frames are live (a tile strictly inside the block).
// The singularity (|sin theta| < 1e-9: the sample takes M / P) is met when a phase comes within 3e-10 of an integer --
// in practice the very first sample of an oscillator that starts at phase 0 -- so the quotients are formed without
// the per-sample select and the comparisons are only OR-ed together on the scalar unit; a wave in which any lane met
// it runs the samples again with the selects (GUARD).  Same operations either way.
template <bool INNER, bool GUARD>
__device__ __forceinline__ unsigned long long saw_dirichlet_rot_body(double sd, double cd, double sn, double cn,
                                                                     const SawConst &k0, const SawRot &rot,
                                                                     double m_over_p, int64_t f0, int64_t n,
                                                                     double (&xb)[kSawT]) {
    unsigned long long any = 0ull;
#pragma unroll
    for (int j = 0; j < kSawT; ++j) {
        if (j) {
            const double s2 = __builtin_fma(sd, rot.cd, cd * rot.sd);
            cd = __builtin_fma(cd, rot.cd, -(sd * rot.sd));
            sd = s2;
            const double n2 = __builtin_fma(sn, rot.cm, cn * rot.sm);
            cn = __builtin_fma(cn, rot.cm, -(sn * rot.sm));
            sn = n2;
        }
        double blit = pgx::pgx_div_fast1(sn, k0.P * sd);
        if (GUARD) {
            if (fabs(sd) < 1e-9) blit = m_over_p;
        } else {
            any |= __ballot(fabs(sd) < 1e-9);
        }
        xb[j] = (INNER || f0 + j < n) ? (blit - k0.invP) : 0.0;
    }
    return any;
}
template <bool INNER = false>
__device__ __forceinline__ void saw_dirichlet_rot(double ph0, const SawConst &k0, const SawRot &rot, double m_over_p,
                                                  int64_t f0, int64_t n, double (&xb)[kSawT]) {
    double sd, cd, sn, cn;
    const double theta = kPi * ph0;
    pgx::pgx_sincos_bounded(theta, sd, cd);
    pgx::pgx_sincos_bounded(k0.m * theta, sn, cn);
    if (saw_dirichlet_rot_body<INNER, false>(sd, cd, sn, cn, k0, rot, m_over_p, f0, n, xb))
        saw_dirichlet_rot_body<INNER, true>(sd, cd, sn, cn, k0, rot, m_over_p, f0, n, xb);
}
// the phase of the thread's sample j (np.mod(phase, 1.0)), j wave-divergent: for the carried state only
// (the thread's local phase sums are re-added rather than indexed: an indexed register array moves to LDS)
__device__ __forceinline__ double saw_phase_at(double phase0, double chunk_base, double inc, int j) {
    double l = 0.0;
#pragma unroll
    for (int q = 0; q < kSawT; ++q) l = (q <= j) ? l + inc : l;
    return pgx::pgx_mod1(phase0 + (chunk_base + l));
}

// STREAMS = false: scalar frequency / amplitude / M (every voice-bank and SuperSaw launch): the
// per-voice constants are hoisted out of the sample loops.
// NW = 4: 256 threads, 2048-frame tiles (banks of oscillators).  NW = 8: 512 threads, 4096-frame tiles --
// the same wave values folded in the same order, i.e. bit-identical output with half of the
// dependent tile steps, for graphs with a few oscillators where the chain length is what costs.
//
// SEG: one long stream (scalar parameters) over several workgroups.  The two carries of a tile are its running
// phase sum and the integrator level; both chains are replayed exactly as the single workgroup would run them,
// so the output is the same bit for bit.  SEG = 1 (reduce): workgroup (inst, s) replays the phase chain up to
// its first tile -- every full tile adds the same per-wave sums -- and records each wave's zero-state integrator
// response (what block_scan_scalar_affine_wide folds).  k_blitsaw_chain then folds those responses in order, once
// (one wave per oscillator: cn = lam_wave * cn + t, 8 steps per tile -- every segment replaying its own prefix
// made the apply pass quadratic in the stream length: 890 us of a 990 us SuperSaw window of 2.8 M frames), and
// SEG = 2 (apply) renders its tiles like SEG = 0 from the carries of its segment.  Workspace per oscillator:
// {phase0, y0, wave responses[tiles * NW], integrator carry[tiles], phase-sum carry[nseg]}.
template <bool STREAMS, int NW, int SEG>
__global__ void __launch_bounds__(NW * 64)
k_blitsaw(float *out, int64_t out_stride, int64_t n, int channels, double sr, const pgx_blitsaw_params *params,
          const float *freq, int64_t freq_stride, const float *amp, int64_t amp_stride, const float *mstream,
          int64_t m_stride, double *state, double *ws, int64_t ws_stride, int tiles_per_seg, double *state_backup) {
    static_assert(!(STREAMS && SEG), "segments need scalar parameters");
    __shared__ SawShared sh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int inst = blockIdx.x;
    const int seg = SEG ? blockIdx.y : 0;
    double *wsi = SEG ? ws + (int64_t)inst * ws_stride : nullptr;
    const pgx_blitsaw_params p = params[inst];
    float *ob = out + (int64_t)inst * out_stride;
    const float *fs = (STREAMS && freq) ? freq + (int64_t)inst * freq_stride : nullptr;
    const float *as = (STREAMS && amp) ? amp + (int64_t)inst * amp_stride : nullptr;
    const float *ms = (STREAMS && mstream) ? mstream + (int64_t)inst * m_stride : nullptr;

    const double phase0 = (SEG == 2) ? wsi[0] : state[inst * 2 + 0];
    double carry_sum = 0.0;                 // running np.cumsum(phase_inc) at the tile start
    double carry_y = (SEG == 2) ? wsi[1] : state[inst * 2 + 1];   // leaky integrator output y[n-1]
    const SawConst k0 = saw_const(p.freq, sr, p.m, false, 0.0);
    const SawRot rot = saw_rot(k0);
    const double m_over_p = k0.m / k0.P;
    if (SEG == 1 && seg == 0 && tid == 0) {                       // the apply pass must not read what it overwrites
        wsi[0] = phase0;
        wsi[1] = carry_y;
    }
    if (SEG != 2 && state_backup != nullptr && seg == 0 && tid == 0) {   // the caller's snapshot of a speculative render
        state_backup[inst * 2 + 0] = phase0;
        state_backup[inst * 2 + 1] = carry_y;
    }

    // powers of leak for the affine scan
    const double leak = p.leak;
    double lamp[6], lam_wave;
    {
        double l = leak;
#pragma unroll
        for (int s = 1; s < kSawT; s <<= 1) l = l * l;      // leak^T
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            lamp[k] = l;
            l = l * l;
        }
        lam_wave = l;
    }
    const LanePowers lane_pw = lane_powers(lamp, lane);

    double final_phase = 0.0, final_y = 0.0;
    bool have_final = false;

    constexpr int kTile = NW * 64 * kSawT;
    const int first_tile = seg * tiles_per_seg;
    // workspace of one oscillator: {phase0, y0, wave responses[tiles * NW], carry_y[tiles], carry_sum[nseg]}
    const int64_t all_tiles = SEG ? (n + kTile - 1) / kTile : 0;
    double *tile_y = SEG ? wsi + 2 + all_tiles * NW : nullptr;
    double *seg_sum = SEG ? tile_y + all_tiles : nullptr;
    if (SEG == 1 && first_tile > 0) {
        // phase chain of the tiles before this segment: block_excl_sum_wide's carry, one add per group of waves
        double run = 0.0;
#pragma unroll
        for (int j = 0; j < kSawT; ++j) run = run + k0.inc;
        const double inc = wave_incl_sum_dpp(run);          // the scan the tiles themselves use: same wave totals
        if (lane == 63) sh.sum[tid >> 6] = inc;
        __syncthreads();
        double tot[NW / kWaves];
#pragma unroll
        for (int g = 0; g < NW / kWaves; ++g) {
            tot[g] = 0.0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) tot[g] = tot[g] + sh.sum[g * kWaves + w];
        }
        __syncthreads();
        for (int tile = 0; tile < first_tile; ++tile)
#pragma unroll
            for (int g = 0; g < NW / kWaves; ++g) carry_sum = carry_sum + tot[g];
    }
    if (SEG == 1 && tid == 0) seg_sum[seg] = carry_sum;       // the apply pass starts from it
    if (SEG == 2) {
        carry_sum = seg_sum[seg];
        carry_y = tile_y[first_tile];                          // k_blitsaw_chain: the integrator chain, folded once
    }
    UniformPrefix uni{0.0, 0.0};                               // a full tile's phase prefix, less the carry
    if (!STREAMS) {
        double run = 0.0;
#pragma unroll
        for (int j = 0; j < kSawT; ++j) run = run + k0.inc;
        uni = uniform_prefix(run);
    }
    const int64_t seg_begin = SEG ? (int64_t)first_tile * kTile : 0;
    const int64_t seg_end = SEG ? ((seg_begin + (int64_t)tiles_per_seg * kTile < n) ? seg_begin + (int64_t)tiles_per_seg * kTile : n) : n;
    int parity = 0;
    for (int64_t base = seg_begin; base < seg_end; base += kTile, ++parity) {
        // a tile strictly inside the block (every frame live, the block's last frame elsewhere) takes the body without
        // the per-sample bounds selects and state captures (k_supersaw_bank: ~10 of ~75 instructions per sample)
        auto tile = [&](auto inner_tag) {
        constexpr bool INNER = decltype(inner_tag)::value;
        const int64_t f0 = base + (int64_t)tid * kSawT;
        SawConst kc[kSawT];
        // ---- phase increment and inclusive local cumsum (blit_saw_pe.py:188-191) ----
        double loc[kSawT];
        double run = 0.0;
        // (PE-driven frequency / M / amplitude: the thread's values first, unconditionally -- see k_sine_stateful)
        float fs_in[STREAMS ? kSawT : 1], ms_in[STREAMS ? kSawT : 1], as_in[STREAMS ? kSawT : 1];
        if (STREAMS && fs) {
#pragma unroll
            for (int j = 0; j < kSawT; ++j) fs_in[j] = fs[(f0 + j < n) ? f0 + j : n - 1];
        }
        if (STREAMS && ms) {
#pragma unroll
            for (int j = 0; j < kSawT; ++j) ms_in[j] = ms[(f0 + j < n) ? f0 + j : n - 1];
        }
        if (STREAMS && as) {
#pragma unroll
            for (int j = 0; j < kSawT; ++j) as_in[j] = as[(f0 + j < n) ? f0 + j : n - 1];
        }
#pragma unroll
        for (int j = 0; j < kSawT; ++j) {
            const bool live = INNER || (f0 + j < n);
            if (STREAMS) {
                double f = fs ? (live ? (double)fs_in[j] : 0.0) : p.freq;
                kc[j] = saw_const(f, sr, p.m, ms != nullptr, ms ? (live ? (double)ms_in[j] : 1.0) : 0.0);
            } else {
                kc[j] = k0;
            }
            run = run + (live ? kc[j].inc : 0.0);
            loc[j] = run;
        }
        // scalar frequency, every frame of the tile live: all threads add the same increments, no exchange -- and on
        // an inner tile the scan of those increments is the one made before the loop (uni)
        const double chunk_base = (!STREAMS && INNER) ? block_excl_sum_wide_uniform<NW>(uni, carry_sum)
                                  : (!STREAMS && base + kTile <= n)
                                      ? block_excl_sum_wide_uniform<NW>(run, carry_sum)
                                      : block_excl_sum_wide<NW>(run, sh.sum, carry_sum);

        // ---- Dirichlet kernel (blit_saw_pe.py:194-217) ----
        double xb[kSawT];
        auto dirichlet = [&](auto bounded) {
#pragma unroll
            for (int j = 0; j < kSawT; ++j) {
                const double ph = pgx::pgx_mod1(phase0 + (chunk_base + loc[j]));      // np.mod(phase, 1.0)
                const double theta = kPi * ph;
                const double m_theta = kc[j].m * theta;
                const double sin_num = saw_sin_num<decltype(bounded)::value>(m_theta);
                const double sin_den = pgx::pgx_sin_bounded(theta);
                // |P*sin_den| >= 1e-9: a Newton-refined reciprocal (<= 1 ulp) replaces the IEEE division sequence.
                // Evaluated unconditionally and then overridden at the singularity (a select, not a branch: the
                // samples of a thread must stay in one basic block to be interleaved)
                double blit = pgx::pgx_div_fast(sin_num, kc[j].P * sin_den);
                if (fabs(sin_den) < 1e-9) blit = kc[j].m / kc[j].P;
                xb[j] = (INNER || f0 + j < n) ? (blit - kc[j].invP) : 0.0;
                if (!INNER && f0 + j == n - 1) {
                    final_phase = ph;
                    have_final = true;
                }
            }
        };
        if (!STREAMS && rot.usable) {
            saw_dirichlet_rot<INNER>(pgx::pgx_mod1(phase0 + (chunk_base + loc[0])), k0, rot, m_over_p, f0, n, xb);
            if (!INNER && f0 <= n - 1 && n - 1 < f0 + kSawT) {
                final_phase = saw_phase_at(phase0, chunk_base, k0.inc, (int)(n - 1 - f0));
                have_final = true;
            }
        } else if (!STREAMS && k0.m < kSawBoundedM) dirichlet(std::true_type{});
        else dirichlet(std::false_type{});

        // ---- leaky integrator y[n] = x[n] + leak*y[n-1] (blit_saw_pe.py:222-234) ----
        double e = 0.0;
#pragma unroll
        for (int j = 0; j < kSawT; ++j) e = __builtin_fma(leak, e, xb[j]);   // feeds the scan only
        double y = block_scan_scalar_affine_wide1<NW>(e, lamp, lam_wave, lane_pw, sh.aff, parity, carry_y);
        if (SEG == 1) {                      // the wave responses the scan just folded are still in LDS
            if (tid < NW) wsi[2 + (base / kTile) * NW + tid] = sh.aff[(parity & 1) * NW + tid];
            return;
        }

        float yf[kSawT];
#pragma unroll
        for (int j = 0; j < kSawT; ++j) {
            double z = leak * y;
            y = z + xb[j];
            double a = p.amp;
            if (STREAMS && as) a = (INNER || f0 + j < n) ? (double)as_in[STREAMS ? j : 0] : 0.0;
            yf[j] = (float)(y * (2.0 * a));                       // (y * 2) * a: the doubling is exact
            if (!INNER && f0 + j == n - 1) final_y = y;
        }
        store_frames_tiled<kSawT>(ob, f0, n, channels, yf);
    
        };
        if (!STREAMS && base + kTile < n) tile(std::true_type{});
        else tile(std::false_type{});
    }
    if (SEG != 1 && have_final) {
        state[inst * 2 + 0] = final_phase;
        state[inst * 2 + 1] = final_y;
    }
}

// The integrator chain over the recorded wave responses of one oscillator (see k_blitsaw, SEG): the carry on
// entering every tile, in the operation order of the single-workgroup path.  One wave per oscillator; the
// responses come 64 at a time (8 tiles), the next 64 in flight while these are folded; the 64 steps of a chunk
// are straight-line code (steps past the end fold zeros into a carry nobody reads).
template <int NW>
__global__ void __launch_bounds__(64)
k_blitsaw_chain(const pgx_blitsaw_params *params, double *ws, int64_t ws_stride, int64_t tiles) {
    static_assert(NW == 8, "a chunk of 64 responses is 8 whole tiles");
    const int inst = blockIdx.x, lane = threadIdx.x;
    double *wsi = ws + (int64_t)inst * ws_stride;
    const double *resp = wsi + 2;
    double *tile_y = wsi + 2 + tiles * NW;
    double lam = params[inst].leak;
#pragma unroll
    for (int s = 1; s < kSawT; s <<= 1) lam = lam * lam;      // leak^T
#pragma unroll
    for (int k = 0; k < 6; ++k) lam = lam * lam;              // ... ^64: one wave's frames (k_blitsaw's lam_wave)
    double c = wsi[1];
    const int64_t total = tiles * NW;
    // four chunks in flight, each refilled as soon as it is folded and needed three chunks later: a chunk is folded in
    // ~0.2 us, a load takes ~0.7 us.  (Unconditional loads from a clamped index, no register rotation: behind a branch,
    // or ahead of a move of the newest value, the compiler waits for the load at once -- that was 55 of the kernel's
    // 75 us on a 2.8 M-frame stream.  Chunks past the end fold zeros into a carry nobody reads.)
    auto fetch = [&](int64_t at) { return resp[at < total ? at : total - 1]; };
    auto fold = [&](double cur, int64_t i0) {
        double keep = 0.0;                                    // lane k: the carry on entering tile i0/8 + k
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if ((i & 7) == 0) keep = (lane == (i >> 3)) ? c : keep;
            c = __builtin_fma(lam, c, readlane_f64(cur, i));   // block_scan_scalar_affine_wide1's fold
        }
        const int64_t tile = (i0 >> 3) + lane;
        if (lane < 8 && tile < tiles) tile_y[tile] = keep;
    };
    double c0 = fetch(lane), c1 = fetch(64 + lane), c2 = fetch(128 + lane), c3 = fetch(192 + lane);
    for (int64_t i0 = 0; i0 < total; i0 += 256) {
        fold(c0, i0);
        c0 = fetch(i0 + 256 + lane);
        fold(c1, i0 + 64);
        c1 = fetch(i0 + 320 + lane);
        fold(c2, i0 + 128);
        c2 = fetch(i0 + 384 + lane);
        fold(c3, i0 + 192);
        c3 = fetch(i0 + 448 + lane);
    }
}

// ------------------------------------------------------------------------------------------------
// A bank of BiquadPE(BlitSawPE) voices in one launch (the C5 voice: blit_saw_pe.py:150-264 feeding
// biquad_pe.py:383-404).  One 256-thread workgroup per voice, 2048-frame tiles rendered exactly as
// k_blitsaw<false, 4, 0> renders them; the float32 oscillator samples then stay in registers and go through the
// constant-coefficient section the way k_biquad_settled runs it: zero-state response of the thread's 8 frames
// (fused multiply-adds: it only feeds carries), Kogge-Stone scan of the affine maps over the wave with the powers
// A^(8*2^k) from LDS, waves folded in order, then the 8 frames re-run from the scanned carry-in in scipy's
// DF-II-T operation order.  The [voices][frames] oscillator buffer is neither written nor read.  Mono, scalar
// parameters.
struct SawBqShared {
    double sum[kWaves];
    double aff[2 * kWaves];
    V2 tot[2][kWaves];
    M2 pstep[6];        // A^(8 * 2^k)
    M2 pwave;           // A^(8 * 64)
    M2 m16[64], m32[64];
};
__global__ void __launch_bounds__(kWaves * 64)
k_blitsaw_biquad(float *out, int64_t out_stride, int64_t n, double sr, const pgx_blitsaw_params *params,
                 double *saw_state, const double *coef, double *bq_state) {
    constexpr int NW = kWaves;
    constexpr int kTile = NW * 64 * kSawT;
    __shared__ SawBqShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int inst = blockIdx.x;
    const pgx_blitsaw_params p = params[inst];
    float *ob = out + (int64_t)inst * out_stride;
    const double b0 = coef[inst * 5 + 0], b1 = coef[inst * 5 + 1], b2 = coef[inst * 5 + 2];
    const double a1 = coef[inst * 5 + 3], a2 = coef[inst * 5 + 4];
    if (tid == 0) {
        M2 q{-a1, 1.0, -a2, 0.0};
#pragma unroll
        for (int t = 1; t < kSawT; t <<= 1) q = mm(q, q);        // A^8
        for (int k = 0; k < 6; ++k) {
            sh.pstep[k] = q;
            q = mm(q, q);
        }
        sh.pwave = q;
    }
    __syncthreads();
    // per-lane powers: A^(8*lane) (carry-in of the wave -> lane); A^(8*((lane&15)+1)) and A^(8*((lane&31)+1)) for
    // the two cross-row steps of the DPP scan.  The latter two wait in LDS (8 doubles per lane would not fit next
    // to the oscillator's registers)
    M2 mlane = m_identity(), m16 = m_identity(), m32 = m_identity();
    {
        const int a16 = (lane & 15) + 1, a32 = (lane & 31) + 1;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const M2 pk = sh.pstep[k];
            if (lane & (1 << k)) mlane = mm(pk, mlane);
            if (a16 & (1 << k)) m16 = mm(pk, m16);
            if (a32 & (1 << k)) m32 = mm(pk, m32);
        }
    }
    if (wave == 0) {
        sh.m16[lane] = m16;
        sh.m32[lane] = m32;
    }
    const M2 pwave = sh.pwave;
    __syncthreads();

    const double phase0 = saw_state[inst * 2 + 0];
    double carry_sum = 0.0;
    double carry_y = saw_state[inst * 2 + 1];
    V2 carry_z{bq_state[inst * 2 + 0], bq_state[inst * 2 + 1]};
    const SawConst k0 = saw_const(p.freq, sr, p.m, false, 0.0);
    const SawRot rot = saw_rot(k0);
    const double m_over_p = k0.m / k0.P;
    const double leak = p.leak;
    double lamp[6], lam_wave;
    {
        double l = leak;
#pragma unroll
        for (int t = 1; t < kSawT; t <<= 1) l = l * l;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            lamp[k] = l;
            l = l * l;
        }
        lam_wave = l;
    }
    const LanePowers lane_pw = lane_powers(lamp, lane);
    UniformPrefix uni;                                         // a full tile's phase prefix, less the carry
    {
        double run = 0.0;
#pragma unroll
        for (int j = 0; j < kSawT; ++j) run = run + k0.inc;
        uni = uniform_prefix(run);
    }
    int parity = 0;
    for (int64_t base = 0; base < n; base += kTile, ++parity) {
        const int64_t f0 = base + (int64_t)tid * kSawT;
        // the oscillator tile; a tile strictly inside the block takes it without the per-sample bounds selects and
        // state captures (see k_supersaw_bank)
        float xf[kSawT];
        auto oscillator = [&](auto inner_tag) {
        constexpr bool INNER = decltype(inner_tag)::value;
        double loc[kSawT];
        double run = 0.0;
#pragma unroll
        for (int j = 0; j < kSawT; ++j) {
            run = run + ((INNER || f0 + j < n) ? k0.inc : 0.0);
            loc[j] = run;
        }
        const double chunk_base = (base + kTile <= n) ? block_excl_sum_wide_uniform<NW>(uni, carry_sum)
                                                      : block_excl_sum_wide<NW>(run, sh.sum, carry_sum);
        double xb[kSawT];
        double final_phase = 0.0, final_y = 0.0;
        auto dirichlet = [&](auto bounded) {
#pragma unroll
            for (int j = 0; j < kSawT; ++j) {
                const double ph = pgx::pgx_mod1(phase0 + (chunk_base + loc[j]));
                const double theta = kPi * ph;
                const double m_theta = k0.m * theta;
                const double sin_num = saw_sin_num<decltype(bounded)::value>(m_theta);
                const double sin_den = pgx::pgx_sin_bounded(theta);
                double blit = pgx::pgx_div_fast(sin_num, k0.P * sin_den);
                if (fabs(sin_den) < 1e-9) blit = m_over_p;
                xb[j] = (INNER || f0 + j < n) ? (blit - k0.invP) : 0.0;
                if (!INNER && f0 + j == n - 1) final_phase = ph;
            }
        };
        if (rot.usable) {
            saw_dirichlet_rot<INNER>(pgx::pgx_mod1(phase0 + (chunk_base + loc[0])), k0, rot, m_over_p, f0, n, xb);
            if (!INNER && f0 <= n - 1 && n - 1 < f0 + kSawT) final_phase = saw_phase_at(phase0, chunk_base, k0.inc, (int)(n - 1 - f0));
        } else if (k0.m < kSawBoundedM) dirichlet(std::true_type{});
        else dirichlet(std::false_type{});
        double e = 0.0;
#pragma unroll
        for (int j = 0; j < kSawT; ++j) e = __builtin_fma(leak, e, xb[j]);   // feeds the scan only
        double y = block_scan_scalar_affine_wide1<NW>(e, lamp, lam_wave, lane_pw, sh.aff, parity, carry_y);
#pragma unroll
        for (int j = 0; j < kSawT; ++j) {
            double z = leak * y;
            y = z + xb[j];
            xf[j] = (INNER || f0 + j < n) ? (float)(y * (2.0 * p.amp)) : 0.0f;       // BlitSawPE's float32 output: (y * 2) * amp, the doubling exact
            if (!INNER && f0 + j == n - 1) final_y = y;
        }
        if (!INNER && f0 <= n - 1 && n - 1 < f0 + kSawT) {                // the thread that renders the last frame
            saw_state[inst * 2 + 0] = final_phase;
            saw_state[inst * 2 + 1] = final_y;
        }
        };
        // (the inner-tile form is not used here: it makes this kernel 6 % faster on its own and the envelope walk that
        // shares the SIMDs with it in a C5 block 30 % slower -- 0.184 -> 0.215 ms per block)
        oscillator(std::false_type{});
        const bool last = f0 <= n - 1 && n - 1 < f0 + kSawT;      // the thread that renders the last frame

        // ---- the filter section on the 8 frames in registers (biquad_pe.py:383-404) ----
        V2 ez{0.0, 0.0};
        const double na1 = -a1, na2 = -a2;
#pragma unroll
        for (int j = 0; j < kSawT; ++j) {
            const double x = (double)xf[j];
            const double yy = __builtin_fma(b0, x, ez.x);
            ez.x = __builtin_fma(na1, yy, __builtin_fma(b1, x, ez.y));
            ez.y = __builtin_fma(na2, yy, b2 * x);
        }
        // inclusive scan over the wave on DPP moves, as k_biquad_settled does it
        ez = mv_add_fma(sh.pstep[0], dpp_v2<0x111, 0xf>(ez), ez);
        ez = mv_add_fma(sh.pstep[1], dpp_v2<0x112, 0xf>(ez), ez);
        ez = mv_add_fma(sh.pstep[2], dpp_v2<0x114, 0xf>(ez), ez);
        ez = mv_add_fma(sh.pstep[3], dpp_v2<0x118, 0xf>(ez), ez);
        ez = mv_add_fma(sh.m16[lane], dpp_v2<0x142, 0xa>(ez), ez);
        ez = mv_add_fma(sh.m32[lane], dpp_v2<0x143, 0xc>(ez), ez);
        if (lane == 63) sh.tot[parity & 1][wave] = ez;
        __syncthreads();
        V2 cw = carry_z, fold = carry_z;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            fold = mv_add_fma(pwave, fold, sh.tot[parity & 1][w]);
            if (w + 1 == wave) cw = fold;                         // this wave's carry-in
        }
        carry_z = fold;
        const V2 ex = dpp_v2<0x138, 0xf>(ez);                     // the lane before, 0 for lane 0
        const V2 zin = mv_add_fma(mlane, cw, ex);
        V2 z = zin;
        float yf[kSawT];
#pragma unroll
        for (int j = 0; j < kSawT; ++j) {                         // scipy lfilter DF-II-T operation order
            const double x = (double)xf[j];
            const double yy = z.x + b0 * x;
            const double z0 = (z.y + b1 * x) - a1 * yy;
            z.y = b2 * x - a2 * yy;
            z.x = z0;
            yf[j] = (float)yy;
        }
        store_frames<kSawT>(ob, f0, n, 1, 0, yf);
        if (last) {                                               // the new filter state: up to the last frame only
            z = zin;
            for (int j = 0; j <= (int)(n - 1 - f0); ++j) {
                const double x = (double)xf[j];
                const double yy = z.x + b0 * x;
                const double z0 = (z.y + b1 * x) - a1 * yy;
                z.y = b2 * x - a2 * yy;
                z.x = z0;
            }
            bq_state[inst * 2 + 0] = z.x;
            bq_state[inst * 2 + 1] = z.y;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// A bank of SuperSawPEs, voices summed on chip (super_saw_pe.py:304-316 on top of blit_saw_pe.py:150-264).
// One 512-thread workgroup per SuperSaw instance.  A 4096-frame tile is rendered voice after voice exactly as
// k_blitsaw<false, 8, 0> renders it (same wave values, same folds: the float32 samples of every oscillator are
// that kernel's, bit for bit); instead of being stored, each voice is added, in voice order, to a float64
// accumulator the thread keeps for its 8 frames -- k_supersaw_sum's arithmetic -- and the tile leaves the chip
// once, as float32(acc * amplitude).  The [instances x voices][frames] intermediate never exists.
// The per-voice carries (phase sum, integrator level) live in LDS between tiles.  Scalar parameters only.
constexpr int kSsMaxVoices = 16;
template <int NW>
struct SsShared {
    double sum[NW];
    double aff[2 * NW];
    double carry_sum[kSsMaxVoices];
    double carry_y[kSsMaxVoices];
    double rot[kSsMaxVoices][4];          // per voice: sin / cos of pi*inc and of M*pi*inc (saw_rot)
    int rot_ok[kSsMaxVoices];
    double kc[kSsMaxVoices][6];           // per voice: inc, M, P, 1/P, M/P, phase0 -- made once, not per tile
    double lam[kSsMaxVoices][8];          // per voice: leak^(8*2^k), k = 0..5, leak^(8*64)
    // per voice and thread, made once per launch (inner tiles: every frame live, so a thread's 8 increments, their
    // wave scan and the folds over the waves are the same numbers tile after tile -- ~70 instructions per voice and
    // tile of the ~600), and per voice and lane the powers of the leak the integrator scan multiplies with
    double amp2[kSsMaxVoices];            // 2 * amplitude (the table slot a never-read run8 used to occupy)
    double tot[kSsMaxVoices];
    double offset[kSsMaxVoices][NW * 64];
    double lane_pw[kSsMaxVoices][64][3];
};
// NW = 4: 2048-frame tiles, two workgroups per CU (512 instances fill the chip in one round and one workgroup's
// barrier waits overlap the other's arithmetic); NW = 8: 4096-frame tiles for fewer instances.  Same bits.
// What a workgroup needs per voice before its first tile and what depends on the parameters only: the Dirichlet
// constants, the rotation sines, the powers of the leak, and per thread / lane the prefix offsets and scan powers.
// ss_make_tables computes them (as every launch did); k_supersaw_tables leaves them in HBM once per bank so that
// the workgroups of later launches -- four per instance in the time-segmented form -- only load them (26 KB, L2).
// Per voice: rot[4], rot_ok, kc[0..4], lam[8] (lam[7] = leak), 2 * amp, tot, offset[256], lane_pw[64][3].
constexpr int kSsTabDoubles = 4 + 1 + 5 + 8 + 2 + 256 + 192;

template <int NW>
__device__ __forceinline__ void ss_make_tables(SsShared<NW> &sh, const pgx_blitsaw_params *pv, int nv, double sr) {
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < nv) {
        const pgx_blitsaw_params pt = pv[tid];
        const SawConst kt = saw_const(pt.freq, sr, pt.m, false, 0.0);
        const SawRot r = saw_rot(kt);
        sh.rot[tid][0] = r.sd; sh.rot[tid][1] = r.cd; sh.rot[tid][2] = r.sm; sh.rot[tid][3] = r.cm;
        sh.rot_ok[tid] = r.usable ? 1 : 0;
        sh.kc[tid][0] = kt.inc; sh.kc[tid][1] = kt.m; sh.kc[tid][2] = kt.P; sh.kc[tid][3] = kt.invP;
        sh.kc[tid][4] = kt.m / kt.P;
        double l = pt.leak;
#pragma unroll
        for (int t = 1; t < kSawT; t <<= 1) l = l * l;      // leak^T
        for (int k = 0; k < 7; ++k) {
            sh.lam[tid][k] = l;
            l = l * l;
        }
        sh.lam[tid][7] = pt.leak;
        sh.amp2[tid] = 2.0 * pt.amp;
    }
    __syncthreads();
    for (int v = 0; v < nv; ++v) {                          // the per-thread tables (all threads, every voice)
        const double inc = sh.kc[v][0];
        double run = 0.0;
#pragma unroll
        for (int j = 0; j < kSawT; ++j) run = run + inc;
        const UniformPrefix u = uniform_prefix(run);
        sh.offset[v][tid] = u.offset;
        if (tid == 0) sh.tot[v] = u.tot;
        if (tid < 64) {
            double lamp[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) lamp[k] = sh.lam[v][k];
            const LanePowers lp = lane_powers(lamp, lane);
            sh.lane_pw[v][lane][0] = lp.lane; sh.lane_pw[v][lane][1] = lp.p16; sh.lane_pw[v][lane][2] = lp.p32;
        }
    }
}

template <int NW>
__device__ __forceinline__ void ss_load_tables(SsShared<NW> &sh, const double *tab, int nv) {
    // all of a batch's loads first (unconditional, clamped index), then the stores: a load inside the branchy store
    // loop is waited for right behind its issue, one memory latency per entry (seven in a row: ~10 us)
    constexpr int U = 8;
    const int total = nv * kSsTabDoubles;
    for (int base = 0; base < total; base += U * NW * 64) {
        double val[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * NW * 64 + (int)threadIdx.x;
            val[u] = tab[idx < total ? idx : total - 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * NW * 64 + (int)threadIdx.x;
            if (idx >= total) continue;
            const int v = idx / kSsTabDoubles, o = idx - v * kSsTabDoubles;
            const double x = val[u];
            if (o < 4) sh.rot[v][o] = x;
            else if (o < 5) sh.rot_ok[v] = x != 0.0 ? 1 : 0;
            else if (o < 10) sh.kc[v][o - 5] = x;
            else if (o < 18) sh.lam[v][o - 10] = x;
            else if (o < 19) sh.amp2[v] = x;
            else if (o < 20) sh.tot[v] = x;
            else if (o < 276) {
#pragma unroll
                for (int rep = 0; rep < NW / kWaves; ++rep) sh.offset[v][(o - 20) + rep * 256] = x;   // periodic in the wave group
            } else {
                const int q = o - 276;
                sh.lane_pw[v][q / 3][q % 3] = x;
            }
        }
    }
}

__global__ void __launch_bounds__(256)
k_supersaw_tables(double *tables, int nv, double sr, const pgx_blitsaw_params *params) {
    __shared__ SsShared<4> sh;
    const int inst = blockIdx.x, tid = threadIdx.x;
    ss_make_tables<4>(sh, params + (int64_t)inst * nv, nv, sr);
    __syncthreads();
    double *tab = tables + (int64_t)inst * nv * kSsTabDoubles;
    for (int idx = tid; idx < nv * kSsTabDoubles; idx += 256) {
        const int v = idx / kSsTabDoubles, o = idx - v * kSsTabDoubles;
        double x;
        if (o < 4) x = sh.rot[v][o];
        else if (o < 5) x = (double)sh.rot_ok[v];
        else if (o < 10) x = sh.kc[v][o - 5];
        else if (o < 18) x = sh.lam[v][o - 10];
        else if (o < 19) x = sh.amp2[v];
        else if (o < 20) x = sh.tot[v];
        else if (o < 276) x = sh.offset[v][o - 20];
        else {
            const int q = o - 276;
            x = sh.lane_pw[v][q / 3][q % 3];
        }
        tab[idx] = x;
    }
}

// Time segments (gridDim.y > 1: a few instances have to fill the chip -- a rank's share of a sharded mix): workgroup
// (instance, s) renders the tiles [s * seg_tiles, (s + 1) * seg_tiles).  What it needs on entering its first tile is,
// per voice, the running phase sum (the same additions the lone workgroup makes tile after tile: replayed, a few
// scalar adds) and the integrator level.  The latter has a closed form for a scalar-frequency oscillator with odd M:
// the DC-free BLIT is (2/P) * sum_{k=1..(M-1)/2} cos(2 pi k phase) (Dirichlet kernel), each harmonic passes the
// leaky integrator 1 / (1 - leak z^-1) with a known complex gain, so the steady-state level at a frame with phase ph is
//     yss(ph) = (2/P) * sum_k Re[ e^(j 2 pi k ph) / (1 - leak e^(-j 2 pi k inc)) ]
// and y - yss decays exactly like leak^frames: y[s0 - 1] = yss(ph[s0 - 1]) + leak^s0 * (y[b - 1] - yss(ph[b - 1]))
// from the carried state at the block start b.  (M - 1)/2 <= a few hundred terms, spread over the workgroup: ~1 % of
// a segment's work; agreement with the recurrence ~1e-14, far below the float32 rounding of the output.
__device__ __forceinline__ void saw_steady_terms(double ph_a, double ph_b, double inc, double leak, int K, int first,
                                                 int stride, double &sum_a, double &sum_b) {
    // lane terms k = first, first + stride, ...: the three unit vectors e^(j 2 pi k x) are evaluated once and then
    // turned by e^(j 2 pi stride x) (a handful of steps: ~1e-16 each) instead of three sincos per term
    double a = 0.0, b = 0.0;
    if (first > K) {
        sum_a = sum_b = 0.0;
        return;
    }
    const double kk = (double)first, st = (double)stride;
    // six evaluations through one copy of the routine (a loop the compiler must not unroll): this code runs once per
    // workgroup, straight from a cold instruction cache -- its size, not its arithmetic, is what it costs
    double arg[6] = {kk * inc, kk * ph_a, kk * ph_b, st * inc, st * ph_a, st * ph_b}, sn[6], cs[6];
#pragma unroll 1
    for (int i = 0; i < 6; ++i) pgx::pgx_sincos_bounded((2.0 * kPi) * arg[i], sn[i], cs[i]);
    double s0 = sn[0], c0 = cs[0], s1 = sn[1], c1 = cs[1], s2 = sn[2], c2 = cs[2];
    const double rs0 = sn[3], rc0 = cs[3], rs1 = sn[4], rc1 = cs[4], rs2 = sn[5], rc2 = cs[5];
    for (int k = first; k <= K; k += stride) {
        const double dr = 1.0 - leak * c0, di = leak * s0;      // 1 - leak e^(-ja) = dr + j di
        const double inv = 1.0 / (dr * dr + di * di);
        a += (c1 * dr + s1 * di) * inv;
        b += (c2 * dr + s2 * di) * inv;
        double t;
        t = __builtin_fma(s0, rc0, c0 * rs0); c0 = __builtin_fma(c0, rc0, -(s0 * rs0)); s0 = t;
        t = __builtin_fma(s1, rc1, c1 * rs1); c1 = __builtin_fma(c1, rc1, -(s1 * rs1)); s1 = t;
        t = __builtin_fma(s2, rc2, c2 * rs2); c2 = __builtin_fma(c2, rc2, -(s2 * rs2)); s2 = t;
    }
    sum_a = a;
    sum_b = b;
}

#ifndef PGX_SS_STAMP
#define PGX_SS_STAMP(i)            /* tools/microbench/ss_phases.hip defines it to record wall_clock64() */
#endif
template <int NW>
__global__ void __launch_bounds__(NW * 64)
k_supersaw_bank(float *out, int64_t out_stride, int nv, int64_t n, int channels, double sr,
                const pgx_blitsaw_params *params, const double *state, double *state_out, const double *amp_scalar,
                int seg_tiles, const double *tables) {
    constexpr int kTile = NW * 64 * kSawT;
    __shared__ SsShared<NW> sh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int inst = blockIdx.x;
    const pgx_blitsaw_params *pv = params + (int64_t)inst * nv;
    const double *sv = state + (int64_t)inst * nv * 2;
    double *sv_out = state_out + (int64_t)inst * nv * 2;
    const int64_t tile_first = (int64_t)blockIdx.y * seg_tiles;
    const int64_t frame_first = tile_first * kTile;
    int64_t frame_end = frame_first + (int64_t)seg_tiles * kTile;
    if (frame_end > n) frame_end = n;
    if (frame_first >= n) return;
    float *ob = out + (int64_t)inst * out_stride;
    const double g = amp_scalar[inst];
    PGX_SS_STAMP(0);
    // the carried state is requested before the tables so that the two memory round trips overlap
    const double st_phase = sv[(tid < nv ? tid : 0) * 2 + 0], st_level = sv[(tid < nv ? tid : 0) * 2 + 1];
    if (tables) ss_load_tables<NW>(sh, tables + (int64_t)inst * nv * kSsTabDoubles, nv);
    else ss_make_tables<NW>(sh, pv, nv, sr);
    if (tid < nv) {
        sh.carry_sum[tid] = 0.0;
        sh.carry_y[tid] = st_level;
        sh.kc[tid][5] = st_phase;                           // (the state is rewritten only after the last tile)
    }
    __syncthreads();
    PGX_SS_STAMP(1);
    if (tile_first > 0) {
        // entering a later segment: phase sums by replaying the lone workgroup's per-tile additions, integrator
        // levels from the closed form (comment above the kernel).  One wave per voice, the harmonics over its lanes:
        // the voices' sums run side by side and meet no barrier.
        const int wave = tid >> 6;
        for (int v = wave; v < nv; v += NW) {
            double cs = 0.0;
            for (int64_t t = 0; t < tile_first * (NW / kWaves); ++t) cs = cs + sh.tot[v];
            const double ph_b = sh.kc[v][5];
            const double ph_a = pgx::pgx_mod1(ph_b + cs);
            const int K = ((int)sh.kc[v][1] - 1) / 2;
            const double leak = sh.lam[v][7];
            double pa, pb;
            saw_steady_terms(ph_a, ph_b, sh.kc[v][0], leak, K, lane + 1, 64, pa, pb);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                pa += __shfl_xor(pa, o);
                pb += __shfl_xor(pb, o);
            }
            if (lane == 0) {
                const double scale = 2.0 * sh.kc[v][3];                       // 2 / P
                // leak^frame_first, frame_first = tile_first tiles of NW * 512 frames: from leak^512 by squarings
                double per_tile = sh.lam[v][6];
#pragma unroll
                for (int t = 1; t < NW; t <<= 1) per_tile = per_tile * per_tile;
                double decay = 1.0;
                for (int64_t e = tile_first; e > 0; e >>= 1) {
                    if (e & 1) decay = decay * per_tile;
                    per_tile = per_tile * per_tile;
                }
                sh.carry_sum[v] = cs;
                sh.carry_y[v] = __builtin_fma(decay, sh.carry_y[v] - scale * pb, scale * pa);
            }
        }
        __syncthreads();
    }
    PGX_SS_STAMP(2);
    int parity = 0;
    for (int64_t base = frame_first; base < frame_end; base += kTile) {
        PGX_SS_STAMP(3 + (int)((base - frame_first) / kTile));
        const int64_t f0 = base + (int64_t)tid * kSawT;
        double acc[kSawT];
#pragma unroll
        for (int j = 0; j < kSawT; ++j) acc[j] = 0.0;
        // Every tile runs the body without per-sample bounds tests: frames past the end of the block are rendered like
        // the others (their phases and levels only reach lanes that hold no live frame, and the stores are bounded),
        // so what the live frames get is the same arithmetic as ever.  LAST = the tile that holds the block's last
        // frame: the thread that renders it also captures the carried state.  (A separate bounds-testing body for that
        // tile cost 5 us more than a plain one -- its code was cold in the instruction cache -- and in the
        // time-segmented form the workgroups that own it are the critical path.)
        auto voices = [&](auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr bool INNER = true;
#pragma unroll 1
        for (int v = 0; v < nv; ++v, ++parity) {
            // the voice's constants come from LDS (made once per launch: per tile they were a third of the work; leak and
            // amplitude too: a global load per voice and tile sat at the head of every voice's chain)
            SawConst k0;
            k0.inc = sh.kc[v][0]; k0.m = sh.kc[v][1]; k0.P = sh.kc[v][2]; k0.invP = sh.kc[v][3];
            const double phase0 = sh.kc[v][5];
            const double leak = sh.lam[v][7];
            const double amp2 = sh.amp2[v];
            double lamp[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) lamp[k] = sh.lam[v][k];
            const LanePowers lane_pw{sh.lane_pw[v][lane][0], sh.lane_pw[v][lane][1], sh.lane_pw[v][lane][2]};
            const double lam_wave = sh.lam[v][6];
            double carry_sum = sh.carry_sum[v], carry_y = sh.carry_y[v];
            double loc[kSawT];
            double chunk_base;
            // the thread's increments, their scan and the folds are this voice's table entries
            if (!sh.rot_ok[v]) {                               // (only the per-sample Dirichlet forms read loc[1..])
                double run = 0.0;
#pragma unroll
                for (int j = 0; j < kSawT; ++j) {
                    run = run + k0.inc;
                    loc[j] = run;
                }
            } else {
                loc[0] = 0.0 + k0.inc;
            }
            chunk_base = block_excl_sum_wide_uniform<NW>(UniformPrefix{sh.offset[v][tid], sh.tot[v]}, carry_sum);
            double xb[kSawT];
            double final_phase = 0.0, final_y = 0.0;
            const double m_over_p = sh.kc[v][4];
            auto dirichlet = [&](auto bounded) {
#pragma unroll
                for (int j = 0; j < kSawT; ++j) {
                    const double ph = pgx::pgx_mod1(phase0 + (chunk_base + loc[j]));
                    const double theta = kPi * ph;
                    const double m_theta = k0.m * theta;
                    const double sin_num = saw_sin_num<decltype(bounded)::value>(m_theta);
                    const double sin_den = pgx::pgx_sin_bounded(theta);
                    double blit = pgx::pgx_div_fast(sin_num, k0.P * sin_den);   // a select, not a branch (k_blitsaw)
                    if (fabs(sin_den) < 1e-9) blit = m_over_p;
                    xb[j] = blit - k0.invP;
                    if (LAST && f0 + j == n - 1) final_phase = ph;
                }
            };
            if (sh.rot_ok[v]) {
                const SawRot rot{sh.rot[v][0], sh.rot[v][1], sh.rot[v][2], sh.rot[v][3], true};
                saw_dirichlet_rot<INNER>(pgx::pgx_mod1(phase0 + (chunk_base + loc[0])), k0, rot, m_over_p, f0, n, xb);
                if (LAST && f0 <= n - 1 && n - 1 < f0 + kSawT) final_phase = saw_phase_at(phase0, chunk_base, k0.inc, (int)(n - 1 - f0));
            } else if (k0.m < kSawBoundedM) dirichlet(std::true_type{});
            else dirichlet(std::false_type{});
            double e = 0.0;
#pragma unroll
            for (int j = 0; j < kSawT; ++j) e = __builtin_fma(leak, e, xb[j]);   // feeds the scan only
            double y = block_scan_scalar_affine_wide1<NW>(e, lamp, lam_wave, lane_pw, sh.aff, parity, carry_y);
#pragma unroll
            for (int j = 0; j < kSawT; ++j) {
                double z = leak * y;
                y = z + xb[j];
                acc[j] += (double)(float)(y * amp2);           // (y * 2) * amp: doubling is exact, so 2 * amp first
                if (LAST && f0 + j == n - 1) final_y = y;
            }
            if (LAST && f0 <= n - 1 && n - 1 < f0 + kSawT) {     // the thread that renders the last frame
                sv_out[v * 2 + 0] = final_phase;
                sv_out[v * 2 + 1] = final_y;
            }
            if (tid == 0) {                                 // every thread holds the same carries
                sh.carry_sum[v] = carry_sum;
                sh.carry_y[v] = carry_y;
            }
        }
        };
        if (base + kTile < n) voices(std::false_type{});
        else voices(std::true_type{});
        float yf[kSawT];
#pragma unroll
        for (int j = 0; j < kSawT; ++j) yf[j] = (float)(acc[j] * g);
        store_frames_tiled<kSawT>(ob, f0, n, channels, yf);
        if (nv == 1) __syncthreads();                       // with more voices the carries written above are
                                                            // read again only after the other voices' barriers
    }
    PGX_SS_STAMP(15);
}

// ------------------------------------------------------------------------------------------------
// The bank with SIXTEEN frames per thread (k_supersaw_wide).  k_supersaw_bank is bound by instruction issue (~480
// float64 instructions per thread, voice and tile of 8 frames: 2 x ~30 for the two anchor sincos, ~60 to fetch the
// voice's constants, ~100 for the integrator scan, ~22 + 7 per frame for the Dirichlet quotient and the replay);
// with 16 frames per thread the per-thread part is paid half as often: ~340 per 8 frames.
// Scalar frequency, odd M (the rule sr / (2 f) always gives one), 0 <= f <= sr / 2 only -- the rotation form of the
// Dirichlet kernel; the host checks that before it picks this kernel.  The phase of frame i of a block is taken as
// frac(phase0 + (i + 1) * inc) directly, not from the running sums k_blitsaw forms: the two agree to ~1e-12 (the
// rounding of a sum of ~1e4), the outputs to ~1e-9 of peak -- this kernel is held to a tolerance (<= 1e-6 of peak
// against the voices rendered one by one, 1e-5 against the oracle), not to the bits of k_blitsaw.
// Time segments as in k_supersaw_bank (closed-form integrator level on entering a later segment).
constexpr int kSswT = 16;
// per voice: [0] inc, M, P, 1/P, M/P, leak, 2*amp, [7] 4 sin^2(pi inc / 2); [8] sin/cos(pi inc), sin/cos(M pi inc);
// [12] (leak^16)^(2^k), k = 0..5; [18] leak^(16*64); [19] 2 cos(M pi inc); [20] sin/cos of a 4096-frame tile's advance
// pi*4096*inc and of M times that (the angles reduced exactly before the sincos); [24] lane powers [3][64]
constexpr int kSswTabDoubles = 24 + 3 * 64;
constexpr int kSswLanePw = 24;
template <int NW>
struct SswShared {
    double aff[4 * NW];                   // two images (used alternately) of two chains' wave aggregates
    double carry_y[kSsMaxVoices];
    double phase0[kSsMaxVoices];
};                                        // (the voices' tables and the threads' anchors: dynamic LDS, sized by the voice count)

__global__ void __launch_bounds__(64)
k_supersaw_wide_tables(double *tables, int nv, double sr, const pgx_blitsaw_params *params) {
    const int inst = blockIdx.x, lane = threadIdx.x;
    for (int v = 0; v < nv; ++v) {
        const pgx_blitsaw_params pt = params[(int64_t)inst * nv + v];
        double *tab = tables + ((int64_t)inst * nv + v) * kSswTabDoubles;
        const SawConst kt = saw_const(pt.freq, sr, pt.m, false, 0.0);
        double lamp[6];
        double l = pt.leak;
#pragma unroll
        for (int t = 1; t < kSswT; t <<= 1) l = l * l;          // leak^16
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            lamp[k] = l;
            l = l * l;
        }
        const LanePowers lp = lane_powers(lamp, lane);
        tab[kSswLanePw + lane] = lp.lane;
        tab[kSswLanePw + 64 + lane] = lp.p16;
        tab[kSswLanePw + 128 + lane] = lp.p32;
        if (lane == 0) {
            const SawRot r = saw_rot(kt);
            tab[0] = kt.inc; tab[1] = kt.m; tab[2] = kt.P; tab[3] = kt.invP; tab[4] = kt.m / kt.P;
            tab[5] = pt.leak; tab[6] = 2.0 * pt.amp;
            {                                                   // 4 sin^2(d / 2), d = pi inc: the denominators' difference recurrence
                double sh_, ch_;
                pgx::pgx_sincos_bounded(0.5 * kPi * kt.inc, sh_, ch_);
                tab[7] = 4.0 * sh_ * sh_;
            }
            tab[8] = r.sd; tab[9] = r.cd; tab[10] = r.sm; tab[11] = r.cm;
#pragma unroll
            for (int k = 0; k < 6; ++k) tab[12 + k] = lamp[k];
            tab[18] = l;                                        // leak^(16 * 64)
            tab[19] = 2.0 * r.cm;                              // the numerator's three-term recurrence
            // a tile's advance: 4096 * inc is exact (a power of two), M times it is taken with its rounding error
            const double x = 4096.0 * kt.inc;
            const double pm = kt.m * x, em = __builtin_fma(kt.m, x, -pm);
            double st, ct, stm, ctm;
            pgx::pgx_sincos_bounded(kPi * (x - floor(x)), st, ct);
            pgx::pgx_sincos_bounded(kPi * ((pm - floor(pm)) + em), stm, ctm);
            // (frac drops whole half-turns: an odd number of them flips both signs of a pair, and with M odd the pairs of
            // theta and M theta flip together -- the quotient of the two sines never sees it)
            const bool odd = fmod(floor(x), 2.0) != 0.0, odd_m = fmod(floor(pm), 2.0) != 0.0;
            tab[20] = odd ? -st : st; tab[21] = odd ? -ct : ct;
            tab[22] = odd_m ? -stm : stm; tab[23] = odd_m ? -ctm : ctm;
        }
    }
}

// xb[j] = blit - 1/P for T consecutive frames from the anchors (sd, cd) = sincos(theta), (sn, cn) = sincos(M theta).
// Per frame: the denominator sin(theta_j) by the difference form of its recurrence -- s[j+1] = s[j] + d[j], d[j+1] = d[j] -
// alpha s[j+1] with alpha = 4 sin^2(d/2) and d[0] = cos(theta) sin d - (alpha/2) sin(theta): 2 operations where the
// rotation of (sin, cos) it replaced (end of round 4) takes 4, and as accurate -- 2e-15 after 15 steps for every step angle
// up to pi/2, measured against the rotation's 2e-15 (its absolute error must stay ~1e-15 because the quotient divides
// by it where it is small; the plain three-term form 2 cos d s[j] - s[j-1] loses digits for small d) --, the numerator
// sin(M theta_j) / P by the three-term recurrence
// n[j+1] = 2 cos(M d) n[j] - n[j-1] (1 operation; M d = M pi inc lies within 2 d of pi/2, so the recurrence does not
// amplify its roundings: ~1e-15 after 15 steps), the quotient with one Newton step on the reciprocal and no residual
// correction (2^-48), one reciprocal per pair of frames.  16 instruction slots per frame where saw_dirichlet_rot_body has 30.
// GUARD as there: the singularity test is only OR-ed together, a wave that met it runs the frames again with selects.
template <int T, bool GUARD>
__device__ __forceinline__ unsigned long long saw_rot_frames(double sd, double cd, double sn, double cn, double invP,
                                                             double m_over_p, double rsd, double alpha, double rsm,
                                                             double rcm, double two_cm, double (&xb)[T]) {
    static_assert(T % 2 == 0, "frames are taken in pairs");
    unsigned long long any = 0ull;
    double n_prev = invP * sn;
    double n_cur = invP * __builtin_fma(sn, rcm, cn * rsm);
    double dd = __builtin_fma(cd, rsd, -((0.5 * alpha) * sd));        // sin(theta + d) - sin(theta)
#pragma unroll
    for (int j = 0; j < T; j += 2) {
        // two frames: the denominators sin(theta_j), sin(theta_j+1) by the difference recurrence, the numerators by theirs
        if (j) {
            sd = sd + dd;
            dd = __builtin_fma(-alpha, sd, dd);
        }
        const double sd0 = sd;
        sd = sd + dd;
        dd = __builtin_fma(-alpha, sd, dd);
        const double sd1 = sd;
        double num0, num1;
        if (j == 0) {
            num0 = n_prev;
            num1 = n_cur;
        } else {
            num0 = __builtin_fma(two_cm, n_cur, -n_prev);
            num1 = __builtin_fma(two_cm, num0, -n_cur);
            n_prev = num0;
            n_cur = num1;
        }
        // ONE reciprocal for the pair (v_rcp_f64 is a quarter-rate instruction: four slots): r = 1 / (sd0 sd1) with one
        // Newton step, then 1 / sd0 = r sd1 and 1 / sd1 = r sd0 -- 4.5 slots per frame instead of 7.  |sin| <= 1, so the
        // product is below 1e-9 whenever a factor is: the singularity test on the product has no false negatives (a
        // false positive -- both factors near 3e-5 -- only sends the wave through the guarded pass).
        const double pr = sd0 * sd1;
        double y = __builtin_amdgcn_rcp(pr);
        y = __builtin_fma(__builtin_fma(-pr, y, 1.0), y, y);
        if (GUARD) {
            // (the guarded pass: a wave that met a tiny product; each frame by its own reciprocal, the reference's test)
            double y0 = __builtin_amdgcn_rcp(sd0), y1 = __builtin_amdgcn_rcp(sd1);
            y0 = __builtin_fma(__builtin_fma(-sd0, y0, 1.0), y0, y0);
            y1 = __builtin_fma(__builtin_fma(-sd1, y1, 1.0), y1, y1);
            xb[j] = (fabs(sd0) < 1e-9 ? m_over_p : num0 * y0) - invP;
            xb[j + 1] = (fabs(sd1) < 1e-9 ? m_over_p : num1 * y1) - invP;
        } else {
            any |= __ballot(fabs(pr) < 1e-9);
            xb[j] = __builtin_fma(num0, y * sd1, -invP);           // blit - 1/P in one operation
            xb[j + 1] = __builtin_fma(num1, y * sd0, -invP);
        }
    }
    return any;
}

struct alignas(16) SswAnchor {
    double sd, cd, sn, cn;
};
constexpr int kSswAnchorVoices = 8;        // anchors are kept for banks of up to this many voices per instance (64 KB of LDS)

template <int NW>
__global__ void __launch_bounds__(NW * 64, 2)                    // two waves per SIMD: 512 instances are two workgroups per CU
k_supersaw_wide(float *out, int64_t out_stride, int nv, int64_t n, int channels, const double *state,
                double *state_out, const double *amp_scalar, int seg_tiles, const double *tables, int keep_anchors) {
    constexpr int T = kSswT, kTile = NW * 64 * T;
    __shared__ SswShared<NW> sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char ssw_dynamic[];
    double *tabs = reinterpret_cast<double *>(ssw_dynamic);                     // [nv][kSswTabDoubles]
    SswAnchor *anchors = keep_anchors ? reinterpret_cast<SswAnchor *>(tabs + nv * kSswTabDoubles) : nullptr;   // [nv][threads]
    const int tid = threadIdx.x, lane = tid & 63;
    const int inst = blockIdx.x;
    const int64_t tile_first = (int64_t)blockIdx.y * seg_tiles;
    const int64_t frame_first = tile_first * kTile;
    int64_t frame_end = frame_first + (int64_t)seg_tiles * kTile;
    if (frame_end > n) frame_end = n;
    if (frame_first >= n) return;
    const double *sv = state + (int64_t)inst * nv * 2;
    double *sv_out = state_out + (int64_t)inst * nv * 2;
    float *ob = out + (int64_t)inst * out_stride;
    const double g = amp_scalar[inst];
    PGX_SS_STAMP(0);
    // carried state and tables: all loads first, then the LDS stores
    const int vi = tid < nv ? tid : 0;
    const double st_phase = sv[vi * 2 + 0], st_level = sv[vi * 2 + 1];
    {
        const double *tab = tables + (int64_t)inst * nv * kSswTabDoubles;
        double *dst = tabs;
        const int total = nv * kSswTabDoubles;
        constexpr int U = 4;
        for (int base = 0; base < total; base += U * NW * 64) {
            double val[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * NW * 64 + tid;
                val[u] = tab[idx < total ? idx : total - 1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * NW * 64 + tid;
                if (idx < total) dst[idx] = val[u];
            }
        }
    }
    if (tid < nv) {
        sh.carry_y[tid] = st_level;
        sh.phase0[tid] = st_phase;
    }
    __syncthreads();
    PGX_SS_STAMP(1);
    if (tile_first > 0) {
        // entering a later segment: the integrator level from the closed form (comment above k_supersaw_bank);
        // one wave per voice, the harmonics over its lanes
        const int wave = tid >> 6;
        for (int v = wave; v < nv; v += NW) {
            const double *tb = tabs + v * kSswTabDoubles;
            const double ph_b = sh.phase0[v];
            const double ph_a = pgx::pgx_mod1(ph_b + (double)frame_first * tb[0]);
            const int K = ((int)tb[1] - 1) / 2;
            const double leak = tb[5];
            double pa, pb;
            saw_steady_terms(ph_a, ph_b, tb[0], leak, K, lane + 1, 64, pa, pb);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                pa += __shfl_xor(pa, o);
                pb += __shfl_xor(pb, o);
            }
            if (lane == 0) {
                const double scale = 2.0 * tb[3];                               // 2 / P
                double per_tile = tb[18];                                       // leak^(16 * 64): one wave's frames
#pragma unroll
                for (int t = 1; t < NW; t <<= 1) per_tile = per_tile * per_tile;
                double decay = 1.0;
                for (int64_t e = tile_first; e > 0; e >>= 1) {
                    if (e & 1) decay = decay * per_tile;
                    per_tile = per_tile * per_tile;
                }
                sh.carry_y[v] = __builtin_fma(decay, sh.carry_y[v] - scale * pb, scale * pa);
            }
        }
        __syncthreads();
    }
    PGX_SS_STAMP(2);
    int parity = 0;
    for (int64_t base = frame_first; base < frame_end; base += kTile) {
        PGX_SS_STAMP(3 + (int)((base - frame_first) / kTile));
        const int64_t f0 = base + (int64_t)tid * T;
        const bool owner = f0 <= n - 1 && n - 1 < f0 + T;      // the thread that renders the block's last frame
        double acc[T];
#pragma unroll
        for (int j = 0; j < T; ++j) acc[j] = 0.0;
        // (frames past the end of the block are rendered like the others: their values reach no live frame and the
        // stores are bounded)
        // V voices side by side in one thread: two independent instruction streams for the scheduler (a workgroup per
        // CU is one wave per SIMD: every dependent float64 operation waited out its predecessor's latency) and one
        // barrier for both integrator scans.
        auto voices = [&](auto vtag, int v0) {
            constexpr int V = decltype(vtag)::value;
            double leak[V], amp2[V], xb[V][T], e[V], lam_wave[V], carry_y[V], incl[V];
            LanePowers lane_pw[V];
#pragma unroll
            for (int u = 0; u < V; ++u) {
                const double *tb = tabs + (v0 + u) * kSswTabDoubles;
                const double inc = tb[0], m = tb[1], invP = tb[3], m_over_p = tb[4], two_cm = tb[19];
                leak[u] = tb[5];
                amp2[u] = tb[6];
                const double rsd = tb[8], alpha = tb[7], rsm = tb[10], rcm = tb[11];
                double lamp[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) lamp[k] = tb[12 + k];
                lam_wave[u] = tb[18];
                lane_pw[u] = LanePowers{tb[kSswLanePw + lane], tb[kSswLanePw + 64 + lane], tb[kSswLanePw + 128 + lane]};
                carry_y[u] = sh.carry_y[v0 + u];
                // The anchor -- sin / cos of theta and M theta at the thread's first frame: evaluated for the workgroup's
                // first tile, kept per (voice, thread) in dynamic LDS and turned by a tile's advance for every further
                // tile (16 instruction slots instead of ~60).  anchors == nullptr (more voices than the LDS holds
                // anchors for): evaluated every tile.
                double sd, cd, sn, cn;
                SswAnchor *anc = anchors ? anchors + (v0 + u) * (NW * 64) + tid : nullptr;
                if (PGX_COLD(anc == nullptr || base == frame_first)) {
                    const double ph = pgx::pgx_mod1(sh.phase0[v0 + u] + (double)(f0 + 1) * inc);
                    const double theta = kPi * ph;
                    pgx::pgx_sincos_bounded(theta, sd, cd);
                    pgx::pgx_sincos_bounded(m * theta, sn, cn);
                } else {
                    const SswAnchor a = *anc;
                    const double ts = tb[20], tc = tb[21], tsm = tb[22], tcm = tb[23];
                    sd = __builtin_fma(a.sd, tc, a.cd * ts);
                    cd = __builtin_fma(a.cd, tc, -(a.sd * ts));
                    sn = __builtin_fma(a.sn, tcm, a.cn * tsm);
                    cn = __builtin_fma(a.cn, tcm, -(a.sn * tsm));
                }
                if (anc != nullptr) *anc = SswAnchor{sd, cd, sn, cn};
                const unsigned long long met = saw_rot_frames<T, false>(sd, cd, sn, cn, invP, m_over_p, rsd, alpha, rsm, rcm, two_cm, xb[u]);
                if (PGX_COLD(met != 0ull))
                    saw_rot_frames<T, true>(sd, cd, sn, cn, invP, m_over_p, rsd, alpha, rsm, rcm, two_cm, xb[u]);
                double f = 0.0;
#pragma unroll
                for (int j = 0; j < T; ++j) f = __builtin_fma(leak[u], f, xb[u][j]);   // feeds the scan only
                e[u] = f;
                incl[u] = wave_incl_affine_dpp(f, lamp, lane_pw[u].p16, lane_pw[u].p32);
            }
            // block_scan_scalar_affine_wide1 for V chains behind one barrier
            double *img = sh.aff + (parity & 1) * (2 * NW);
            if (lane == 63) {
#pragma unroll
                for (int u = 0; u < V; ++u) img[u * NW + (tid >> 6)] = incl[u];
            }
            __syncthreads();
            ++parity;
#pragma unroll
            for (int u = 0; u < V; ++u) {
                double cw = carry_y[u], cn = carry_y[u];
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const double t = img[u * NW + w];
                    if (w < (tid >> 6)) cw = __builtin_fma(lam_wave[u], cw, t);
                    cn = __builtin_fma(lam_wave[u], cn, t);
                }
                const double ex = dpp_f64_keep<0x138, 0xf>(0.0, incl[u]);
                double y = __builtin_fma(lane_pw[u].lane, cw, ex);
                const double y_in = y;
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    y = __builtin_fma(leak[u], y, xb[u][j]);       // (fused like the fold above: ~1e-16 per step)
                    // (the voice is not rounded to float32 before it joins the sum, as a BlitSawPE's output would be:
                    // <= 6e-8 of a voice's level each, inside this kernel's tolerance; three operations fewer per frame)
                    acc[j] = __builtin_fma(y, amp2[u], acc[j]);
                }
                if (PGX_COLD(owner)) {                         // (one thread of the block's last tile)
                    const int jn = (int)(n - 1 - f0);
                    double yl = y_in;
#pragma unroll
                    for (int j = 0; j < T; ++j) yl = j <= jn ? __builtin_fma(leak[u], yl, xb[u][j]) : yl;
                    sv_out[(v0 + u) * 2 + 1] = yl;
                }
                if (tid == 0) sh.carry_y[v0 + u] = cn;         // every thread holds the same carry
            }
        };
        // (one voice at a time: two side by side -- one barrier for both scans, two instruction streams -- measured no
        // faster and, with a reciprocal per pair of frames, no longer fit 256 registers)
#pragma unroll 1
        for (int v = 0; v < nv; ++v) voices(std::integral_constant<int, 1>{}, v);
        float yf[T];
#pragma unroll
        for (int j = 0; j < T; ++j) yf[j] = (float)(acc[j] * g);
        store_frames_tiled<T>(ob, f0, n, channels, yf);
        if (nv <= 1) __syncthreads();                           // with more voices the carries written above are read
                                                                // again only after the other voices' barriers
    }
    if (frame_end == n && tid < nv) sv_out[tid * 2 + 0] = pgx::pgx_mod1(sh.phase0[tid] + (double)n * tabs[tid * kSswTabDoubles]);
    PGX_SS_STAMP(15);
}

// ------------------------------------------------------------------------------------------------
// BlitSawPE -> BiquadPE (scalar parameters) with sixteen frames per thread: k_blitsaw_biquad's chain with
// k_supersaw_wide's oscillator (phases as products, numerator by the three-term recurrence, one Newton step) and the
// filter section of the settled biquad kernels -- zero-state pass, DPP scan on the per-voice tables of A^(16 * 2^k)
// (pgx_biquad_tables), then the carried state's contribution (A^j z).x added to the zero-state outputs (two fused
// multiply-adds per frame instead of the recurrence again).  One workgroup of NW waves per voice walks the block;
// the oscillator's float32 rounding is BlitSawPE's ((y * 2) * amp).  Held to a tolerance like k_supersaw_wide
// (<= 1e-6 of peak against the two-launch bank), not to bits.
template <int NW>
struct SawBqWideShared {
    double aff[2 * NW];
    V2 tot[2][NW];
};
// SEG (round 4: a rank's share of C5, banks of up to 256 voices): workgroup (voice, s) renders the tiles of time segment
// s, so that a few voices fill the chip.  What it needs on entering: the oscillator's phase -- a product, as everywhere in
// this kernel -- and integrator level -- the closed form of k_supersaw_wide's segments (saw_steady_terms) -- and the
// filter's state, which has no closed form but is forgotten: the segment starts `warm_tiles` tiles early from a zero filter
// state (the host's settle_frames: every entry of A^W below 2^-90) and emits nothing there.  States are read from one
// buffer and written to another (the workgroup of the last segment writes while those of the others may still read).
template <int NW, bool GAIN, bool SEG = false>
__global__ void __launch_bounds__(NW * 64)
k_blitsaw_biquad_wide(float *out, int64_t out_stride, int64_t n, const double *saw_tables, double *saw_state,
                      const double *coef, const double *bq_tables, double *bq_state, const float *gain,
                      int64_t gain_stride, double *saw_state_out = nullptr, double *bq_state_out = nullptr,
                      int seg_tiles = 0, int warm_tiles = 0) {
    constexpr int T = kSswT, kTile = NW * 64 * T;
    static_assert(kSswT == kBqT, "the filter tables are made for 16 frames per thread");
    __shared__ SawBqWideShared<NW> sh;
    __shared__ double sh_entry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int inst = blockIdx.x;
    double *saw_out = SEG ? saw_state_out : saw_state, *bq_out = SEG ? bq_state_out : bq_state;
    int64_t base_first = 0, emit_from = 0, base_end = n;
    if (SEG) {
        const int64_t t0 = (int64_t)blockIdx.y * seg_tiles;
        emit_from = t0 * kTile;
        if (emit_from >= n) return;
        base_first = (t0 > warm_tiles ? t0 - warm_tiles : 0) * kTile;
        base_end = emit_from + (int64_t)seg_tiles * kTile;
        if (base_end > n) base_end = n;
    }
    float *ob = out + (int64_t)inst * out_stride;
    // uniform loads: the voice's constants live in scalar registers
    const double *st = saw_tables + (int64_t)inst * kSswTabDoubles;
    const double inc = st[0], m = st[1], invP = st[3], m_over_p = st[4], leak = st[5], amp2 = st[6];
    const double rsd = st[8], alpha = st[7], rsm = st[10], rcm = st[11], two_cm = st[19];
    double lamp[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) lamp[k] = st[12 + k];
    const double lam_wave = st[18];
    const LanePowers lane_pw{st[kSswLanePw + lane], st[kSswLanePw + 64 + lane], st[kSswLanePw + 128 + lane]};
    const double phase0 = saw_state[inst * 2 + 0];
    double carry_y = saw_state[inst * 2 + 1];
    const double b0 = coef[inst * 5 + 0], b1 = coef[inst * 5 + 1], b2 = coef[inst * 5 + 2];
    const double na1 = -coef[inst * 5 + 3], na2 = -coef[inst * 5 + 4];
    const double *tb = bq_tables + (int64_t)inst * kBqTableDoubles;
    M2 pstep[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pstep[k] = load_m2(tb + 4 * k);
    const M2 pwave = load_m2(tb + 24);
    const M2 mlane = load_m2(tb + 28 + 4 * lane);
    const M2 m16 = load_m2(tb + 28 + 4 * ((lane & 15) + 1));
    const M2 m32 = load_m2(tb + 28 + 4 * ((lane & 31) + 1));
    const double *rows = tb + kBqRowsAt;
    V2 carry_z{bq_state[inst * 2 + 0], bq_state[inst * 2 + 1]};
    if (SEG && base_first > 0) {
        // the integrator level on entering the segment (comment above k_supersaw_bank): one wave, the harmonics over
        // its lanes; the filter starts from rest, `warm_tiles` tiles before the first frame it emits
        carry_z = V2{0.0, 0.0};
        if (wave == 0) {
            const double ph_a = pgx::pgx_mod1(phase0 + (double)base_first * inc);
            const int K = ((int)m - 1) / 2;
            double pa, pb;
            saw_steady_terms(ph_a, phase0, inc, leak, K, lane + 1, 64, pa, pb);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                pa += __shfl_xor(pa, o);
                pb += __shfl_xor(pb, o);
            }
            if (lane == 0) {
                const double scale = 2.0 * invP;
                double per_tile = lam_wave;                               // leak^(16 * 64): one wave's frames
#pragma unroll
                for (int t = 1; t < NW; t <<= 1) per_tile = per_tile * per_tile;
                double decay = 1.0;
                for (int64_t e = base_first / kTile; e > 0; e >>= 1) {
                    if (e & 1) decay = decay * per_tile;
                    per_tile = per_tile * per_tile;
                }
                sh_entry = __builtin_fma(decay, carry_y - scale * pb, scale * pa);
            }
        }
        __syncthreads();
        carry_y = sh_entry;
    }
    const double tile_s = st[20], tile_c = st[21], tile_sm = st[22], tile_cm = st[23];
    double a_sd = 0.0, a_cd = 1.0, a_sn = 0.0, a_cn = 1.0;
    int parity = 0;
    const float *gb = GAIN ? gain + (int64_t)inst * gain_stride : nullptr;     // (GAIN: its own instantiation -- as a run-time
                                                                                // option it cost the plain kernel 19 VGPRs and 8 %)
    for (int64_t base = base_first; base < base_end; base += kTile, ++parity) {
        const int64_t f0 = base + (int64_t)tid * T;
        const bool emit = !SEG || base >= emit_from;               // (a segment's warm-up tiles only move the states)
        // the voice's gain (GainPE(x, gain=<PE>): float32 x float32) for these frames, asked for first
        float gv[GAIN ? T : 1];
        if (GAIN) {
            if (PGX_HOT(f0 + T <= n && aligned16(gb + f0))) {
#pragma unroll
                for (int j = 0; j < T; j += 4) {
                    const float4 q = *reinterpret_cast<const float4 *>(gb + f0 + j);
                    gv[j] = q.x; gv[j + 1] = q.y; gv[j + 2] = q.z; gv[j + 3] = q.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < T; ++j) gv[j] = f0 + j < n ? gb[f0 + j] : 0.0f;
            }
        }
        // ---- the oscillator (frames past the block's end are rendered like the others and never stored)
        // The anchor -- sin / cos of theta and M theta at the thread's first frame -- is evaluated for the first tile and
        // turned by a tile's advance (table: angles reduced exactly) for every further one: 8 operations instead of two
        // sincos (~60); a block's 12 tiles add ~1e-15.
        if (PGX_COLD(base == base_first)) {
            const double ph = pgx::pgx_mod1(phase0 + (double)(f0 + 1) * inc);
            const double theta = kPi * ph;
            pgx::pgx_sincos_bounded(theta, a_sd, a_cd);
            pgx::pgx_sincos_bounded(m * theta, a_sn, a_cn);
        } else {
            const double s2 = __builtin_fma(a_sd, tile_c, a_cd * tile_s);
            a_cd = __builtin_fma(a_cd, tile_c, -(a_sd * tile_s));
            a_sd = s2;
            const double n2 = __builtin_fma(a_sn, tile_cm, a_cn * tile_sm);
            a_cn = __builtin_fma(a_cn, tile_cm, -(a_sn * tile_sm));
            a_sn = n2;
        }
        const double sd = a_sd, cd = a_cd, sn = a_sn, cn = a_cn;
        double xb[T];
        const unsigned long long met = saw_rot_frames<T, false>(sd, cd, sn, cn, invP, m_over_p, rsd, alpha, rsm, rcm, two_cm, xb);
        if (PGX_COLD(met != 0ull))
            saw_rot_frames<T, true>(sd, cd, sn, cn, invP, m_over_p, rsd, alpha, rsm, rcm, two_cm, xb);
        double e = 0.0;
#pragma unroll
        for (int j = 0; j < T; ++j) e = __builtin_fma(leak, e, xb[j]);           // feeds the scan only
        double y = block_scan_scalar_affine_wide1<NW>(e, lamp, lam_wave, lane_pw, sh.aff, parity, carry_y);
        const double y_in = y;
        // ---- BlitSawPE's float32 sample, and the filter's zero-state pass over the 16 frames
        V2 ez{0.0, 0.0};
        double yz[T];
#pragma unroll
        for (int j = 0; j < T; ++j) {
            y = __builtin_fma(leak, y, xb[j]);
            const double x = (double)(float)(y * amp2);           // (y * 2) * amp, the doubling exact
            const double yy = __builtin_fma(b0, x, ez.x);
            ez.x = __builtin_fma(na1, yy, __builtin_fma(b1, x, ez.y));
            ez.y = __builtin_fma(na2, yy, b2 * x);
            yz[j] = yy;
        }
        const V2 ez_own = ez;
        ez = mv_add_fma(pstep[0], dpp_v2<0x111, 0xf>(ez), ez);
        ez = mv_add_fma(pstep[1], dpp_v2<0x112, 0xf>(ez), ez);
        ez = mv_add_fma(pstep[2], dpp_v2<0x114, 0xf>(ez), ez);
        ez = mv_add_fma(pstep[3], dpp_v2<0x118, 0xf>(ez), ez);
        ez = mv_add_fma(m16, dpp_v2<0x142, 0xa>(ez), ez);
        ez = mv_add_fma(m32, dpp_v2<0x143, 0xc>(ez), ez);
        V2 *tot = sh.tot[parity & 1];
        if (lane == 63) tot[wave] = ez;
        __syncthreads();
        V2 cw = carry_z, fold = carry_z;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            fold = mv_add_fma(pwave, fold, tot[w]);
            if (w + 1 == wave) cw = fold;                         // this wave's carry-in
        }
        carry_z = fold;
        const V2 ex = dpp_v2<0x138, 0xf>(ez);                     // the lane before, 0 for lane 0
        const V2 zin = mv_add_fma(mlane, cw, ex);
        if (PGX_HOT(emit)) {
            float yf[T];
#pragma unroll
            for (int j = 0; j < T; ++j)
                yf[j] = (float)__builtin_fma(rows[2 * j], zin.x, __builtin_fma(rows[2 * j + 1], zin.y, yz[j]));
            if (GAIN) {
#pragma unroll
                for (int j = 0; j < T; ++j) yf[j] = yf[j] * gv[GAIN ? j : 0];   // gain_pe.py:104-119: the float32 product
            }
            store_frames<T>(ob, f0, n, 1, 0, yf);
        }
        if (PGX_COLD(f0 <= n - 1 && n - 1 < f0 + T)) {            // the thread that renders the block's last frame:
            const int jn = (int)(n - 1 - f0);                     // the states after it, by the literal recurrences
            double yl = y_in;
            V2 z = zin;
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const double yn = __builtin_fma(leak, yl, xb[j]);
                const double x = (double)(float)(yn * amp2);
                const double yy = z.x + b0 * x;
                const double z0 = (z.y + b1 * x) + na1 * yy;
                const double z1 = b2 * x + na2 * yy;
                if (j <= jn) {
                    yl = yn;
                    z = V2{z0, z1};
                }
            }
            saw_out[inst * 2 + 1] = yl;
            bq_out[inst * 2 + 0] = z.x;
            bq_out[inst * 2 + 1] = z.y;
        }
        (void)ez_own;
    }
    if (tid == 0 && (!SEG || base_end == n)) saw_out[inst * 2 + 0] = pgx::pgx_mod1(phase0 + (double)n * inc);
}

// ------------------------------------------------------------------------------------------------
// A bank of BlitSawPE -> BiquadPE [-> GainPE(gain=<envelope>)] voices under a MixPE, MIXED ON CHIP (round 4): the
// [voices][frames] layer between the voices and the mix (mix_pe.py:91-94 over gain_pe.py:104-119) does not exist.
//
// The unit of work is a (voice, tile) pair that depends on no other pair: workgroup (tile t, group g) renders the group's
// voices one after the other over the SAME 4096 frames -- k_blitsaw_biquad_wide's body: oscillator, integrator scan,
// float32 rounding, filter, float32 product with the envelope -- and adds each to a float64 accumulator its threads keep
// for their 16 frames; the tile leaves the chip once, as one row of partial sums per group, and k_mix_partials adds the
// rows in group order (a fixed order: the result does not depend on the launch).  What a pair needs on entering:
//   * the oscillator's phase: a product, as everywhere in these kernels; the threads' anchors sincos(theta), sincos(M theta)
//     are the tile's first anchor (k_voice_tile_entries) turned by the thread's offset (a per-voice table of 256 angles
//     reduced exactly: k_voice_tile_pack) -- 12 operations where two sincos take ~120;
//   * the integrator's level: saw_steady_terms' closed form from the level the block began with (k_voice_tile_entries:
//     one wave per pair, ahead of the tiles);
//   * the filter's state: forgotten.  Tile t starts `warm` frames (a multiple of 16: whole threads) before the first frame it
//     emits, from rest; `warm` is the bank's settle horizon (every entry of A^warm below 2^-90), a few hundred frames
//     where k_blitsaw_biquad_wide<SEG> spends whole tiles.  Tile 0 starts at the block's first frame from the carried state.
// Tiles advance by emit = 4096 - warm frames.  The thread that renders the block's last frame writes the carried states
// (to the *_out buffers: other workgroups still read the *_in ones).  <= 1e-6 of peak against k_blitsaw_biquad_wide
// followed by k_gain_mix_batch, like every closed-form entry in this file.
// ------------------------------------------------------------------------------------------------
constexpr int kVtEntryDoubles = 8;       // per (voice, tile): (2/P) sum_a, leak^frames, the first thread's anchor (4), (2/P) sum_b, spare
// One voice's constants, packed (k_voice_tile_pack) so that a workgroup brings them into LDS with three 4 KB LDS-DMA
// instructions per thread-row: [0] the 24 scalars of pgx_supersaw_wide_tables, [24] b0 b1 b2 a1 a2, [32] the first 28 doubles
// of pgx_biquad_tables (A^(16 2^k), A^(16 64)), [64] leak^(16 k) for k = 0..63 (the scans' per-lane powers are entries
// lane, (lane & 15) + 1 and (lane & 31) + 1 of it), [128] the first rows of A^j, j = 0..15, [256] A^(16 j) for j = 0..63,
// [512] per thread l: sincos(pi 16 l inc), sincos(M pi 16 l inc) -- the turn from the tile's first anchor to thread l's.
constexpr int kVtPack = 1536;            // doubles per voice: 12 KB
constexpr int kVtLds = kVtPack + kVtEntryDoubles;

__global__ void __launch_bounds__(256)
k_voice_tile_pack(double *pack, const double *saw_tables, const double *coef, const double *bq_tables) {
    const int inst = blockIdx.x, l = threadIdx.x;
    const double *st = saw_tables + (int64_t)inst * kSswTabDoubles;
    const double *tb = bq_tables + (int64_t)inst * kBqTableDoubles;
    double *dst = pack + (int64_t)inst * kVtPack;
    if (l < 24) dst[l] = st[l];
    if (l < 8) dst[24 + l] = l < 5 ? coef[inst * 5 + l] : 0.0;
    if (l < 32) dst[32 + l] = l < 28 ? tb[l] : 0.0;
    if (l < 64) dst[64 + l] = st[kSswLanePw + l];
    if (l < 32) dst[128 + l] = tb[kBqRowsAt + l];
    if (l >= 160) dst[l] = 0.0;
    dst[256 + l] = tb[28 + l];
    const double inc = st[0], m = st[1];
    // 16 l inc and M 16 l inc as (product, its rounding error); whole half-turns are dropped with their sign
    const double k1 = 16.0 * (double)l, km = m * k1;                    // integers: exact
    double out4[4];
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
        const double k = i ? km : k1;
        const double p = k * inc, e = __builtin_fma(k, inc, -p);
        const double fl = floor(p);
        double sn, cs;
        pgx::pgx_sincos_bounded(kPi * ((p - fl) + e), sn, cs);
        const bool odd = fmod(fl, 2.0) != 0.0;
        out4[2 * i] = odd ? -sn : sn;
        out4[2 * i + 1] = odd ? -cs : cs;
    }
    double *r = dst + 512 + 4 * l;
    r[0] = out4[0]; r[1] = out4[1]; r[2] = out4[2]; r[3] = out4[3];
}

// One workgroup per voice, all tiles of the block: the harmonics over the threads (saw_steady_terms' sum; the unit
// vectors of consecutive tiles by a fixed turn -- a tile's advance, reduced exactly), then one thread per tile for the
// anchors.  advance: the block begins that many frames after the one `saw_state` is the start of (its phase by the same
// expression k_voice_tiles leaves it with: the entries of the NEXT block can be made while this one is rendered).
constexpr int kVtEntryTiles = 15;                           // tiles per pass (+ the sum at the block's start: 16 rows)
__device__ __forceinline__ void voice_tile_entries_body(double (*part)[256], int v, double *entries, const double *pack,
                                                        const double *saw_state, int64_t advance, int ntiles,
                                                        int emit_frames, int warm) {
    constexpr int TT = kVtEntryTiles;
    const int tid = threadIdx.x;
    const double *st = pack + (int64_t)v * kVtPack;
    const double inc = st[0], m = st[1], invP = st[3], leak = st[5];
    double phase0 = saw_state[v * 2 + 0];
    if (advance > 0) phase0 = pgx::pgx_mod1(phase0 + (double)advance * inc);
    const int K = ((int)m - 1) / 2;
    const double scale = 2.0 * invP;
    double *dst = entries + (int64_t)v * ntiles * kVtEntryDoubles;
    // a tile's advance in phase: emit_frames * inc as (product, rounding error), whole turns dropped
    const double ef = (double)emit_frames;
    const double adv_p = ef * inc, adv_e = __builtin_fma(ef, inc, -adv_p);
    const double adv = (adv_p - floor(adv_p)) + adv_e;
    for (int t0 = 1; t0 < ntiles; t0 += TT) {               // tiles t0 .. t0 + TT - 1 (tile 0 enters with the carried level)
        const int64_t base0 = (int64_t)t0 * emit_frames - warm;
        const double ph_a = pgx::pgx_mod1(phase0 + (double)base0 * inc);
        double sum_a[TT], sum_b = 0.0;
#pragma unroll
        for (int u = 0; u < TT; ++u) sum_a[u] = 0.0;
        for (int k = tid + 1; k <= K; k += 256) {
            const double kk = (double)k;
            double arg[4] = {kk * inc, kk * ph_a, kk * phase0, kk * adv}, sn[4], cs[4];
#pragma unroll 1
            for (int i = 0; i < 4; ++i) pgx::pgx_sincos_bounded((2.0 * kPi) * (arg[i] - floor(arg[i])), sn[i], cs[i]);
            const double dr = 1.0 - leak * cs[0], di = leak * sn[0];      // 1 - leak e^(-ja) = dr + j di
            const double inv = 1.0 / (dr * dr + di * di);
            const double gr = dr * inv, gi = di * inv;
            sum_b += cs[2] * gr + sn[2] * gi;
            double c1 = cs[1], s1 = sn[1];
#pragma unroll
            for (int u = 0; u < TT; ++u) {
                sum_a[u] += c1 * gr + s1 * gi;
                const double t2 = __builtin_fma(s1, cs[3], c1 * sn[3]);
                c1 = __builtin_fma(c1, cs[3], -(s1 * sn[3]));
                s1 = t2;
            }
        }
#pragma unroll
        for (int u = 0; u < TT; ++u) part[u][tid] = sum_a[u];
        part[TT][tid] = sum_b;
        __syncthreads();
        // thread (row, piece): 16 of the row's 256 terms, then the 16 pieces of a row over their 16 lanes
        const int row = tid >> 4, piece = tid & 15;
        double x = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) x += part[row][piece * 16 + i];
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) x += __shfl_xor(x, o);
        __syncthreads();
        if (piece == 0) part[0][row] = x;
        __syncthreads();
        const int t = t0 + tid;
        if (tid < TT && t < ntiles) {
            const int64_t base = (int64_t)t * emit_frames - warm;
            double decay = 1.0, q = leak;
            for (int64_t e = base; e > 0; e >>= 1) {
                if (e & 1) decay = decay * q;
                q = q * q;
            }
            double *d = dst + (int64_t)t * kVtEntryDoubles;
            d[0] = scale * part[0][tid]; d[1] = decay; d[6] = scale * part[0][TT]; d[7] = 0.0;
        }
        __syncthreads();
    }
    // the first thread's anchor of every tile: k_blitsaw_biquad_wide's first-tile evaluation for the frame base + 1
    for (int t = tid; t < ntiles; t += 256) {
        const int64_t base = t == 0 ? 0 : (int64_t)t * emit_frames - warm;
        const double ph = pgx::pgx_mod1(phase0 + (double)(base + 1) * inc);
        const double theta = kPi * ph;
        double sd, cd, sn, cn;
        pgx::pgx_sincos_bounded(theta, sd, cd);
        pgx::pgx_sincos_bounded(m * theta, sn, cn);
        double *d = dst + (int64_t)t * kVtEntryDoubles;
        d[2] = sd; d[3] = cd; d[4] = sn; d[5] = cn;
        if (t == 0) { d[0] = 0.0; d[1] = 1.0; d[6] = 0.0; d[7] = 0.0; }
    }
}

__global__ void __launch_bounds__(256)
k_voice_tile_entries(double *entries, const double *pack, const double *saw_state, int64_t advance, int ntiles,
                     int emit_frames, int warm) {
    __shared__ double part[kVtEntryTiles + 1][256];
    voice_tile_entries_body(part, blockIdx.x, entries, pack, saw_state, advance, ntiles, emit_frames, warm);
}

#ifndef PGX_VT_WAVES
#define PGX_VT_WAVES 3                  // waves per SIMD the kernel is compiled for
#endif
typedef __attribute__((address_space(1))) const void *vt_global_ptr;
typedef __attribute__((address_space(3))) void *vt_lds_ptr;

__device__ __forceinline__ double vt_uniform(double x) {          // a value every lane holds -> scalar registers
    const unsigned long long b = __double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

template <int NW, bool GAIN>
__global__ void __launch_bounds__(NW * 64, PGX_VT_WAVES)
k_voice_tiles(double *__restrict__ partial, int64_t partial_stride, int64_t n, int nvoices, int groups, int ntiles,
              int emit_frames, int warm, const double *__restrict__ pack, const double *__restrict__ saw_state,
              double *__restrict__ saw_out, const double *__restrict__ bq_state, double *__restrict__ bq_out,
              const double *__restrict__ entries, const float *__restrict__ gain, int64_t gain_stride) {
    constexpr int T = kSswT;
    static_assert(NW * 64 == 256, "the packed tables hold one turn per thread of a 256-thread workgroup");
    __shared__ __attribute__((aligned(16))) double cbuf[2][kVtLds];
    __shared__ SawBqWideShared<NW> sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroup -> (group, tile).  Consecutive workgroups go to consecutive XCDs: with the groups a multiple of 8 the
    // workgroups of one group -- the only readers of its voices' tables -- all land on XCD (group mod 8), whose L2 then
    // holds an eighth of the bank's tables
    int g, t;
    if ((groups & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = groups >> 3;
        g = (slot % per) * 8 + xcd;
        t = slot / per;
    } else {
        g = blockIdx.x % groups;
        t = blockIdx.x / groups;
    }
    const int64_t base = t == 0 ? 0 : (int64_t)t * emit_frames - warm;
    const int64_t f0 = base + (int64_t)tid * T;
    const int64_t emit_lo = (int64_t)t * emit_frames;
    const int64_t emit_hi = emit_lo + emit_frames < n ? emit_lo + emit_frames : n;
    const bool owner = emit_lo <= n - 1 && n - 1 < emit_hi && f0 <= n - 1 && n - 1 < f0 + T;
    double acc[T];
#pragma unroll
    for (int j = 0; j < T; ++j) acc[j] = 0.0;
    const int v_first = (int)((int64_t)g * nvoices / groups), v_end = (int)((int64_t)(g + 1) * nvoices / groups);
    // a voice's constants: global memory -> LDS without passing through registers, one voice ahead of their use
    auto stage = [&](int v, int which) {
        const char *src = reinterpret_cast<const char *>(pack + (int64_t)v * kVtPack) + tid * 16;
        double *dst = cbuf[which] + wave * 128;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            __builtin_amdgcn_global_load_lds((vt_global_ptr)(src + q * 4096), (vt_lds_ptr)(dst + q * 512), 16, 0, 0);
        if (tid < 4)
            __builtin_amdgcn_global_load_lds(
                (vt_global_ptr)(reinterpret_cast<const char *>(entries + ((int64_t)v * ntiles + t) * kVtEntryDoubles) + tid * 16),
                (vt_lds_ptr)(cbuf[which] + kVtPack), 16, 0, 0);
    };
    if (v_first < v_end) stage(v_first, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int parity = 0;
#pragma unroll 1
    for (int v = v_first; v < v_end; ++v, ++parity) {
        const double *cb = cbuf[parity & 1];
        const double level0 = saw_state[v * 2 + 1];
        // ---- the oscillator: the tile's first anchor turned by this thread's offset
        const double *en = cb + kVtPack;
        const double *rt = cb + 512 + 4 * tid;
        const double r_s = rt[0], r_c = rt[1], r_sm = rt[2], r_cm = rt[3];
        const double sd = __builtin_fma(en[2], r_c, en[3] * r_s), cd = __builtin_fma(en[3], r_c, -(en[2] * r_s));
        const double sn = __builtin_fma(en[4], r_cm, en[5] * r_sm), cn = __builtin_fma(en[5], r_cm, -(en[4] * r_sm));
        const double inc = cb[0], invP = vt_uniform(cb[3]), m_over_p = cb[4], leak = vt_uniform(cb[5]), amp2 = cb[6];
        const double rsd = vt_uniform(cb[8]), alpha = vt_uniform(cb[7]), rsm = cb[10], rcm = cb[11], two_cm = vt_uniform(cb[19]);
        double xb[T];
        const unsigned long long met = saw_rot_frames<T, false>(sd, cd, sn, cn, invP, m_over_p, rsd, alpha, rsm, rcm, two_cm, xb);
        if (PGX_COLD(met != 0ull))
            saw_rot_frames<T, true>(sd, cd, sn, cn, invP, m_over_p, rsd, alpha, rsm, rcm, two_cm, xb);
        double e = 0.0;
#pragma unroll
        for (int j = 0; j < T; ++j) e = __builtin_fma(leak, e, xb[j]);           // feeds the scan only
        double lamp[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) lamp[k] = cb[12 + k];
        const double lam_wave = cb[18];
        const LanePowers lane_pw{cb[64 + lane], cb[64 + (lane & 15) + 1], cb[64 + (lane & 31) + 1]};
        double carry_y = t == 0 ? level0 : __builtin_fma(en[1], level0 - en[6], en[0]);
        double y = block_scan_scalar_affine_wide1<NW>(e, lamp, lam_wave, lane_pw, sh.aff, parity, carry_y);
        const double y_in = y;
        // ---- behind the first barrier every wave has left the voice before: its buffer takes the next voice's constants;
        // this voice's gains (GainPE(x, gain=<PE>): float32 x float32) are asked for, a zero-state pass ahead of their use
        if (v + 1 < v_end) stage(v + 1, (parity + 1) & 1);
        float gv[GAIN ? T : 1];
        if (GAIN) {
            const float *gb = gain + (int64_t)v * gain_stride;
            if (PGX_HOT(f0 + T <= n && aligned16(gb + f0))) {
#pragma unroll
                for (int j = 0; j < T; j += 4) {
                    const float4 q = *reinterpret_cast<const float4 *>(gb + f0 + j);
                    gv[j] = q.x; gv[j + 1] = q.y; gv[j + 2] = q.z; gv[j + 3] = q.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < T; ++j) gv[j] = f0 + j < n ? gb[f0 + j] : 0.0f;
            }
        }
        // ---- the filter's zero-state pass over the 16 frames.  The oscillator's sample (y * 2) * amp goes in unrounded
        // (BlitSawPE rounds it to float32, 6e-8 of the voice's level: three operations per frame; k_supersaw_wide does the
        // same), so its scale folds into b0, b1, b2
        const double b0 = vt_uniform(cb[24] * amp2), b1 = vt_uniform(cb[25] * amp2), b2 = vt_uniform(cb[26] * amp2);
        const double na1 = vt_uniform(-cb[27]), na2 = vt_uniform(-cb[28]);
        V2 ez{0.0, 0.0};
        double yz[T];
#pragma unroll
        for (int j = 0; j < T; ++j) {
            y = __builtin_fma(leak, y, xb[j]);
            const double yy = __builtin_fma(b0, y, ez.x);
            ez.x = __builtin_fma(na1, yy, __builtin_fma(b1, y, ez.y));
            ez.y = __builtin_fma(na2, yy, b2 * y);
            yz[j] = yy;
        }
        // (the scan's matrices are asked for here, not at the top of the zero-state pass: 56 registers the pass has no room for)
        __builtin_amdgcn_sched_barrier(0);
        const double *tb = cb + 32;
        ez = mv_add_fma(load_m2(tb + 0), dpp_v2<0x111, 0xf>(ez), ez);
        ez = mv_add_fma(load_m2(tb + 4), dpp_v2<0x112, 0xf>(ez), ez);
        ez = mv_add_fma(load_m2(tb + 8), dpp_v2<0x114, 0xf>(ez), ez);
        ez = mv_add_fma(load_m2(tb + 12), dpp_v2<0x118, 0xf>(ez), ez);
        ez = mv_add_fma(load_m2(cb + 256 + 4 * ((lane & 15) + 1)), dpp_v2<0x142, 0xa>(ez), ez);
        ez = mv_add_fma(load_m2(cb + 256 + 4 * ((lane & 31) + 1)), dpp_v2<0x143, 0xc>(ez), ez);
        V2 *tot = sh.tot[parity & 1];
        if (lane == 63) tot[wave] = ez;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the next voice's constants have landed: LDS-DMA counts as a load)
        __syncthreads();
        const V2 carry_z = t == 0 ? V2{bq_state[v * 2 + 0], bq_state[v * 2 + 1]} : V2{0.0, 0.0};
        const M2 pwave = load_m2(tb + 24);
        V2 cw = carry_z, fold = carry_z;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            fold = mv_add_fma(pwave, fold, tot[w]);
            if (w + 1 == wave) cw = fold;                         // this wave's carry-in
        }
        const V2 ex = dpp_v2<0x138, 0xf>(ez);                     // the lane before, 0 for lane 0
        const V2 zin = mv_add_fma(load_m2(cb + 256 + 4 * lane), cw, ex);
        // the carried state's contribution (A^j z).x by its own two-operation recurrence (the rows of A^j from LDS, two
        // fused multiply-adds per frame, are an operation less -- and 32 more values in flight: measured with spills)
        V2 hz = zin;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            float yf = (float)(yz[j] + hz.x);
            const double hx = __builtin_fma(na1, hz.x, hz.y);
            hz.y = na2 * hz.x;
            hz.x = hx;
            if (GAIN) yf = yf * gv[GAIN ? j : 0];                 // gain_pe.py:104-119: the float32 product
            acc[j] += (double)yf;                                 // mix_pe.py:91-94, in float64 until the rows are added
        }
        if (PGX_COLD(owner)) {                                    // the thread that renders the block's last frame:
            const int jn = (int)(n - 1 - f0);                     // the states after it, by the literal recurrences
            double yl = y_in;
            V2 z = zin;
            double xo[T];                                         // (the oscillator's frames again, from the anchors again:
            const double o_sd = __builtin_fma(en[2], rt[1], en[3] * rt[0]), o_cd = __builtin_fma(en[3], rt[1], -(en[2] * rt[0]));   // kept until
            const double o_sn = __builtin_fma(en[4], rt[3], en[5] * rt[2]), o_cn = __builtin_fma(en[5], rt[3], -(en[4] * rt[2]));   // here they cost
            saw_rot_frames<T, true>(o_sd, o_cd, o_sn, o_cn, cb[3], cb[4], cb[8], cb[7], cb[10], cb[11], cb[19], xo);               // every thread 40 registers)
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const double yn = __builtin_fma(leak, yl, xo[j]);
                const double yy = z.x + b0 * yn;
                const double z0 = (z.y + b1 * yn) + na1 * yy;
                const double z1 = b2 * yn + na2 * yy;
                if (j <= jn) {
                    yl = yn;
                    z = V2{z0, z1};
                }
            }
            saw_out[v * 2 + 1] = yl;
            bq_out[v * 2 + 0] = z.x;
            bq_out[v * 2 + 1] = z.y;
        }
        if (PGX_COLD(t == 0 && tid == 0)) saw_out[v * 2 + 0] = pgx::pgx_mod1(saw_state[v * 2 + 0] + (double)n * inc);
    }
    // the group's row of partial sums: this thread's 16 frames are all inside the tile's emitted range or all outside
    // (emit_frames and warm are multiples of 16)
    if (f0 >= emit_lo && f0 < emit_hi) {
        double *row = partial + (int64_t)g * partial_stride + f0;
        if (PGX_HOT(f0 + T <= n)) {
#pragma unroll
            for (int j = 0; j < T; j += 2) *reinterpret_cast<double2 *>(row + j) = double2{acc[j], acc[j + 1]};
        } else {
#pragma unroll
            for (int j = 0; j < T; ++j)
                if (f0 + j < n) row[j] = acc[j];
        }
    }
}

// out[f] = float32(sum over the groups' rows, in group order)
__device__ __forceinline__ void mix_partials_body(int64_t block, float *out, const double *partial, int64_t partial_stride,
                                                  int groups, int64_t n) {
    const int64_t f = block * 256 + threadIdx.x;
    if (f >= n) return;
    constexpr int U = 16;
    double acc = 0.0;
    int g = 0;
    for (; g + U <= groups; g += U) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = partial[(int64_t)(g + u) * partial_stride + f];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (g < groups) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = g + u < groups ? partial[(int64_t)(g + u) * partial_stride + f] : 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    out[f] = (float)acc;
}

__global__ void __launch_bounds__(256)
k_mix_partials(float *out, const double *partial, int64_t partial_stride, int groups, int64_t n) {
    mix_partials_body(blockIdx.x, out, partial, partial_stride, groups, n);
}

// The rows' sum and, in the same launch, what the tiles of the NEXT block of the stream enter with (from the states this
// block's k_voice_tiles has just written): two kernels' worth of independent workgroups, one dispatch -- the entries (7 us
// alone) ride beside the sum instead of behind it on the block's chain.  Workgroups [0, nvoices): entries; the rest: rows.
__global__ void __launch_bounds__(256)
k_mix_partials_entries(float *out, const double *partial, int64_t partial_stride, int groups, int64_t n, int nvoices,
                       double *entries, const double *pack, const double *saw_state, int ntiles, int emit_frames, int warm) {
    __shared__ double part[kVtEntryTiles + 1][256];
    if ((int)blockIdx.x < nvoices)
        voice_tile_entries_body(part, blockIdx.x, entries, pack, saw_state, 0, ntiles, emit_frames, warm);
    else
        mix_partials_body((int64_t)blockIdx.x - nvoices, out, partial, partial_stride, groups, n);
}

// Several workgroups per oscillator pay two launches and the Dirichlet kernel twice: worth it from 3 tiles on.
struct SawPlan {
    int nseg, tiles_per_seg;
    int64_t tiles;
};
SawPlan saw_plan(int batch, int64_t n, bool streams) {
    constexpr int64_t tile = kSawWideWaves * 64 * kSawT;
    SawPlan p{1, 0, (n + tile - 1) / tile};
    static const int seg_max_batch = getenv("PGX_SAW_SEG_MAX_BATCH") ? atoi(getenv("PGX_SAW_SEG_MAX_BATCH")) : 128;
    if (streams || batch >= seg_max_batch || p.tiles < 3 || p.tiles > 65536) return p;
    p.tiles_per_seg = (int)((p.tiles + 255) / 256);
    p.nseg = (int)((p.tiles + p.tiles_per_seg - 1) / p.tiles_per_seg);
    return p;
}

// ================================================================================================
// SinePE, stateful path
// ================================================================================================
constexpr int kSinT = 8;
constexpr int kSinTile = kBlock * kSinT;

__global__ void __launch_bounds__(kBlock)
k_sine_stateful(float *out, int64_t n, int channels, double sr, const pgx_sine_stateful_params *params,
                const float *freq, const float *amp, const float *phase_mod, double *state) {
    __shared__ double lds[kWaves];
    const int tid = threadIdx.x;
    const pgx_sine_stateful_params p = params[0];
    const double two_pi = 2.0 * kPi;
    // sine_pe.py:203-214: first render starts from the scalar phase (0 when phase is a PE)
    const bool inited = state[1] != 0.0;
    const double initial = inited ? state[0] : (p.phase_is_stream ? 0.0 : p.phase);
    double carry_sum = 0.0;
    double final_phase = 0.0;
    bool have_final = false;

    for (int64_t base = 0; base < n; base += kSinTile) {
        const int64_t f0 = base + (int64_t)tid * kSinT;
        double loc[kSinT];
        double run = 0.0;
        // the control values of the thread's frames first, unconditionally (clamped index): behind the bounds test every
        // load is waited for at once, one memory latency per value on a kernel that is one workgroup walking the block
        float f_in[kSinT], pm_in[kSinT], a_in[kSinT];
        if (freq) {
#pragma unroll
            for (int j = 0; j < kSinT; ++j) f_in[j] = freq[(f0 + j < n) ? f0 + j : n - 1];
        }
        if (phase_mod) {
#pragma unroll
            for (int j = 0; j < kSinT; ++j) pm_in[j] = phase_mod[(f0 + j < n) ? f0 + j : n - 1];
        }
        if (amp) {
#pragma unroll
            for (int j = 0; j < kSinT; ++j) a_in[j] = amp[(f0 + j < n) ? f0 + j : n - 1];
        }
#pragma unroll
        for (int j = 0; j < kSinT; ++j) {
            double f = p.freq;
            if (freq) f = (f0 + j < n) ? (double)f_in[j] : 0.0;
            double inc = (f0 + j < n) ? (two_pi * f) / sr : 0.0;
            run = run + inc;
            loc[j] = run;
        }
        double tile_total;
        double off = block_excl_sum(run, lds, tile_total);
        const double chunk_base = carry_sum + off;
        carry_sum = carry_sum + tile_total;

        float yf[kSinT];
#pragma unroll
        for (int j = 0; j < kSinT; ++j) {
            double ph = (chunk_base + loc[j]) + initial;             // cumsum + initial_phase (:217)
            double pm = p.phase;
            if (phase_mod) pm = (f0 + j < n) ? (double)pm_in[j] : 0.0;
            ph = ph + pm;                                            // + phase_mod (:220-223)
            double a = p.amp;
            if (amp) a = (f0 + j < n) ? (double)a_in[j] : 0.0;
            yf[j] = (float)(a * pgx::pgx_sin(ph));
            if (f0 + j == n - 1) {
                final_phase = ph;
                have_final = true;
            }
        }
        store_frames_tiled<kSinT>(out, f0, n, channels, yf);
    }
    if (have_final) {
        state[0] = final_phase;
        state[1] = 1.0;
    }
}

// ================================================================================================
// BiquadPE, time-varying coefficients
// ================================================================================================
constexpr int kBvT = 4;
constexpr int kBvTile = kBlock * kBvT;

struct BvShared {
    M2 wm[kWaves];
    V2 wv[kWaves];
};

// Time-varying 2x2 recurrences (varying biquad, SVF): s' = A_n s + b_n.  Three ways to run a chain:
//   MODE 2  one workgroup walks the whole block (short blocks: one launch)
//   MODE 0  reduce: each workgroup composes the affine map (M, v) of its segment of tiles -> agg
//   MODE 1  apply: each workgroup folds the maps of the segments before it onto the carried state and
//           renders its segment
// so a 44 100-frame block is 2 launches of 44 workgroups instead of 1 workgroup doing 44 tiles in a row.
constexpr int kS2MaxSeg = 128;

struct Scan2Plan {
    int seg_tiles, nseg;
};
inline Scan2Plan scan2_plan(int64_t n, int tile) {
    const int64_t tiles = pgx::ceil_div(n, tile);
    Scan2Plan p{(int)tiles, 1};
    if (tiles <= 2) return p;
    int64_t nseg = tiles < kS2MaxSeg ? tiles : kS2MaxSeg;
    p.seg_tiles = (int)pgx::ceil_div(tiles, nseg);
    p.nseg = (int)pgx::ceil_div(tiles, p.seg_tiles);
    return p;
}

// Per tile: every thread holds the composed map (cm, cv) of its T frames.  Inclusive Kogge-Stone over the
// wave, exchange through LDS.  MODE 0 folds the tile into the segment map (acc_m, acc_v); the other modes
// return the state on entering this thread's frames (from the carried state vector) and advance the carry.
template <int MODE>
__device__ __forceinline__ V2 scan2_tile(BvShared &sh, const M2 &cm, const V2 &cv, V2 &carry, M2 &acc_m, V2 &acc_v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    M2 im = cm;
    V2 iv = cv;
    // inclusive scan of the maps over the wave on DPP (row shifts, then the row joins); (om, ov) = the map of the
    // frames before this lane's, composed first
#define PGX_S2_STEP(CTRL, MASK)                                   \
    {                                                             \
        const M2 om = dpp_m2_identity<CTRL, MASK>(im);            \
        const V2 ov = dpp_v2_zero<CTRL, MASK>(iv);                \
        iv = vadd(mv(im, ov), iv);                                \
        im = mm(im, om);                                          \
    }
    PGX_S2_STEP(0x111, 0xf) PGX_S2_STEP(0x112, 0xf) PGX_S2_STEP(0x114, 0xf) PGX_S2_STEP(0x118, 0xf)
    PGX_S2_STEP(0x142, 0xa) PGX_S2_STEP(0x143, 0xc)
#undef PGX_S2_STEP
    if (lane == 63) {
        sh.wm[wave] = im;
        sh.wv[wave] = iv;
    }
    __syncthreads();
    V2 s{0.0, 0.0};
    if (MODE == 0) {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const M2 tm = sh.wm[w];
            acc_v = vadd(mv(tm, acc_v), sh.wv[w]);
            acc_m = mm(tm, acc_m);
        }
    } else {
        V2 cw = carry, cn = carry;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const M2 tm = sh.wm[w];
            const V2 tv = sh.wv[w];
            if (w < wave) cw = vadd(mv(tm, cw), tv);
            cn = vadd(mv(tm, cn), tv);
        }
        carry = cn;
        // exclusive prefix within the wave applied to the wave carry-in
        const M2 em = dpp_m2_identity<0x138, 0xf>(im);              // wave_shr:1
        const V2 ev = dpp_v2_zero<0x138, 0xf>(iv);
        s = vadd(mv(em, cw), ev);
    }
    __syncthreads();
    return s;
}

// carried state on entering segment `seg`: the maps of the earlier segments applied in order.  The maps are staged
// through LDS with one coalesced load first: read one by one from HBM on the dependent chain, 43 of them cost
// 10 us -- more than the segment itself.
__device__ __forceinline__ V2 scan2_fold(const double *agg, int seg, V2 s) {
    __shared__ double maps[kS2MaxSeg * 6];
    for (int i = threadIdx.x; i < seg * 6; i += kBlock) maps[i] = agg[i];
    __syncthreads();
    for (int j = 0; j < seg; ++j) {
        const double *a = maps + j * 6;
        const M2 m{a[0], a[1], a[2], a[3]};
        s = vadd(mv(m, s), V2{a[4], a[5]});
    }
    __syncthreads();
    return s;
}

// RBJ cookbook coefficients for one sample (biquad_pe.py:217-335), normalised by a0.
__device__ __forceinline__ void rbj(int mode, double f, double q, double A, double sqrtA, double sr, double &b0,
                                    double &b1, double &b2, double &a1, double &a2) {
    const double nyq99 = (sr / 2.0) * 0.99;
    f = f < 1.0 ? 1.0 : (f > nyq99 ? nyq99 : f);              // np.clip(freq, 1.0, nyquist*0.99)
    q = q < 0.01 ? 0.01 : (q > 100.0 ? 100.0 : q);
    // Divisions: a Newton-refined reciprocal (<= 1 ulp) instead of the correctly rounded IEEE sequence -- seven of
    // those were 40 % of this kernel's instructions, and the coefficients feed a result held to 1e-5
    double omega = pgx::pgx_div_fast((2.0 * kPi) * f, sr);
    double sn, cs;
    pgx::pgx_sincos_bounded(omega, sn, cs);                   // omega < pi: branch-free, the samples of a thread interleave
    double alpha = pgx::pgx_div_fast(sn, 2.0 * q);
    double a0;
    switch (mode) {
    case 0:
        b0 = (1.0 - cs) / 2.0; b1 = 1.0 - cs; b2 = (1.0 - cs) / 2.0;
        a0 = 1.0 + alpha; a1 = -2.0 * cs; a2 = 1.0 - alpha;
        break;
    case 1:
        b0 = (1.0 + cs) / 2.0; b1 = -(1.0 + cs); b2 = (1.0 + cs) / 2.0;
        a0 = 1.0 + alpha; a1 = -2.0 * cs; a2 = 1.0 - alpha;
        break;
    case 2:
        b0 = alpha; b1 = 0.0; b2 = -alpha;
        a0 = 1.0 + alpha; a1 = -2.0 * cs; a2 = 1.0 - alpha;
        break;
    case 3:
        b0 = 1.0; b1 = -2.0 * cs; b2 = 1.0;
        a0 = 1.0 + alpha; a1 = -2.0 * cs; a2 = 1.0 - alpha;
        break;
    case 4:
        b0 = 1.0 - alpha; b1 = -2.0 * cs; b2 = 1.0 + alpha;
        a0 = 1.0 + alpha; a1 = -2.0 * cs; a2 = 1.0 - alpha;
        break;
    case 5:
        b0 = 1.0 + alpha * A; b1 = -2.0 * cs; b2 = 1.0 - alpha * A;
        a0 = 1.0 + pgx::pgx_div_fast(alpha, A); a1 = -2.0 * cs; a2 = 1.0 - pgx::pgx_div_fast(alpha, A);
        break;
    case 6:
        b0 = A * (((A + 1.0) - (A - 1.0) * cs) + (2.0 * sqrtA) * alpha);
        b1 = (2.0 * A) * ((A - 1.0) - (A + 1.0) * cs);
        b2 = A * (((A + 1.0) - (A - 1.0) * cs) - (2.0 * sqrtA) * alpha);
        a0 = ((A + 1.0) + (A - 1.0) * cs) + (2.0 * sqrtA) * alpha;
        a1 = -2.0 * ((A - 1.0) + (A + 1.0) * cs);
        a2 = ((A + 1.0) + (A - 1.0) * cs) - (2.0 * sqrtA) * alpha;
        break;
    default:
        b0 = A * (((A + 1.0) + (A - 1.0) * cs) + (2.0 * sqrtA) * alpha);
        b1 = (-2.0 * A) * ((A - 1.0) + (A + 1.0) * cs);
        b2 = A * (((A + 1.0) + (A - 1.0) * cs) - (2.0 * sqrtA) * alpha);
        a0 = ((A + 1.0) - (A - 1.0) * cs) + (2.0 * sqrtA) * alpha;
        a1 = 2.0 * ((A - 1.0) - (A + 1.0) * cs);
        a2 = ((A + 1.0) - (A - 1.0) * cs) - (2.0 * sqrtA) * alpha;
        break;
    }
    const double inv_a0 = pgx::pgx_div_fast(1.0, a0);
    b0 = b0 * inv_a0; b1 = b1 * inv_a0; b2 = b2 * inv_a0; a1 = a1 * inv_a0; a2 = a2 * inv_a0;
}

// State map per sample: (y1,y2) -> (y0,y1) with
// y0 = ((ff - a1*y1) - a2*y2), ff = (b0*x + b1*x1) + b2*x2  (biquad_pe.py:53-55 grouping).
// state[channel] = {x1, x2, y1, y2}; snapshot = its copy taken by the reduce launch; agg[channel][seg] = map.
template <int MODE>
__global__ void __launch_bounds__(kBlock)
k_biquad_varying(float *out, const float *in, int64_t n, int channels, double sr,
                 const pgx_biquad_var_params *params, const float *freq, const float *qs, double A,
                 double sqrtA, double *state, double *snapshot, double *agg, int seg_tiles, int nseg) {
    __shared__ BvShared sh;
    const int tid = threadIdx.x;
    const int seg = blockIdx.x, ch = blockIdx.y;
    const pgx_biquad_var_params p = params[0];
    double *st = state + ch * 4;                    // x1,x2,y1,y2
    const double *init = (MODE == 1) ? snapshot + ch * 4 : st;
    const double sx1 = init[0], sx2 = init[1];
    double *ag = agg + ((int64_t)ch * nseg) * 6;
    V2 carry{init[2], init[3]};                     // (y1, y2)
    if (MODE == 1) carry = scan2_fold(ag, seg, carry);
    if (MODE == 0 && seg == 0 && tid < 4) snapshot[ch * 4 + tid] = st[tid];
    M2 acc_m = m_identity();
    V2 acc_v{0.0, 0.0};
    V2 final_y{0.0, 0.0};
    bool have_final = false;

    const int64_t tile0 = (int64_t)seg * seg_tiles;
    for (int t = 0; t < seg_tiles; ++t) {
        const int64_t base = (tile0 + t) * kBvTile;
        if (base >= n) break;
        const int64_t f0 = base + (int64_t)tid * kBvT;
        double ff[kBvT], ca1[kBvT], ca2[kBvT];
        // x history for the feed-forward part
        double xm1, xm2;
        {
            int64_t i1 = f0 - 1, i2 = f0 - 2;
            xm1 = (i1 >= 0) ? ((i1 < n) ? (double)in[i1 * channels + ch] : 0.0) : (i1 == -1 ? sx1 : sx2);
            xm2 = (i2 >= 0) ? ((i2 < n) ? (double)in[i2 * channels + ch] : 0.0) : (i2 == -1 ? sx1 : sx2);
        }
        M2 cm = m_identity();
        V2 cv{0.0, 0.0};
        // the thread's samples and control values first, unconditionally (clamped index): a load behind the bounds test
        // is waited for at once, and the 3 x T loads of a thread then run one memory latency after the other
        float x_in[kBvT], f_in[kBvT], q_in[kBvT];
#pragma unroll
        for (int j = 0; j < kBvT; ++j) {
            const int64_t at = (f0 + j < n) ? f0 + j : n - 1;
            x_in[j] = in[at * channels + ch];
        }
        if (freq) {
#pragma unroll
            for (int j = 0; j < kBvT; ++j) f_in[j] = freq[(f0 + j < n) ? f0 + j : n - 1];
        }
        if (qs) {
#pragma unroll
            for (int j = 0; j < kBvT; ++j) q_in[j] = qs[(f0 + j < n) ? f0 + j : n - 1];
        }
#pragma unroll
        for (int j = 0; j < kBvT; ++j) {
            const bool live = (f0 + j < n);
            double x = live ? (double)x_in[j] : 0.0;
            double f = p.freq, q = p.q;
            if (freq) f = live ? (double)f_in[j] : 1000.0;
            if (qs) q = live ? (double)q_in[j] : 1.0;
            double b0, b1, b2, a1, a2;
            rbj(p.mode, f, q, A, sqrtA, sr, b0, b1, b2, a1, a2);
            ff[j] = (b0 * x + b1 * xm1) + b2 * xm2;
            ca1[j] = a1;
            ca2[j] = a2;
            xm2 = xm1;
            xm1 = x;
            if (live) {
                // compose: s' = M s + v with M = [[-a1,-a2],[1,0]], v = [ff,0]
                M2 nm;
                nm.a = -a1 * cm.a - a2 * cm.c;
                nm.b = -a1 * cm.b - a2 * cm.d;
                nm.c = cm.a;
                nm.d = cm.b;
                V2 nv;
                nv.x = (ff[j] - a1 * cv.x) - a2 * cv.y;
                nv.y = cv.x;
                cm = nm;
                cv = nv;
            }
        }
        V2 s = scan2_tile<MODE>(sh, cm, cv, carry, acc_m, acc_v);
        if (MODE == 0) continue;

        float yf[kBvT];
#pragma unroll
        for (int j = 0; j < kBvT; ++j) {
            double y0 = (ff[j] - ca1[j] * s.x) - ca2[j] * s.y;
            yf[j] = (float)y0;
            if (f0 + j < n) {
                s.y = s.x;
                s.x = y0;
            }
            if (f0 + j == n - 1) {
                final_y = s;
                have_final = true;
            }
        }
        store_frames<kBvT>(out, f0, n, channels, ch, yf);
    }
    if (MODE == 0) {
        if (tid == 0) {
            double *a = ag + (int64_t)seg * 6;
            a[0] = acc_m.a; a[1] = acc_m.b; a[2] = acc_m.c; a[3] = acc_m.d; a[4] = acc_v.x; a[5] = acc_v.y;
        }
        return;
    }
    if (have_final) {
        // x1 = x[n-1]; x2 = x[n-2] (or the previous x1 when n == 1), numba kernel semantics
        double nx1 = (double)in[(n - 1) * channels + ch];
        double nx2 = (n >= 2) ? (double)in[(n - 2) * channels + ch] : sx1;
        st[0] = nx1;
        st[1] = nx2;
        st[2] = final_y.x;
        st[3] = final_y.y;
    }
}

// ================================================================================================
// PeriodicGate with PE-driven frequency / duty / phase: FunctionGenPE's stateful rectangle path
// (function_gen_pe.py:157-193): base = mod(phase0 + [0, cumsum(f/sr)[:-1]], 1); gate = mod(base + ph, 1) < duty.
// One workgroup: exclusive prefix sum of the increments, like the stateful sine.
// state = { carried phase }.
// ================================================================================================
constexpr int kGateT = 8;
constexpr int kGateTile = kBlock * kGateT;

__global__ void __launch_bounds__(kBlock)
k_gate_stateful(float *out, int64_t n, double sr, double freq_scalar, double duty_scalar, double phase_scalar,
                const float *freq, const float *duty, const float *phase, double *state) {
    __shared__ double lds[kWaves];
    const int tid = threadIdx.x;
    const double phase0 = state[0];
    double carry_sum = 0.0;
    for (int64_t base = 0; base < n; base += kGateTile) {
        const int64_t f0 = base + (int64_t)tid * kGateT;
        double loc[kGateT];
        double run = 0.0;
        // control values first, unconditionally (see k_sine_stateful)
        float f_in[kGateT], ph_in[kGateT], d_in[kGateT];
        if (freq) {
#pragma unroll
            for (int j = 0; j < kGateT; ++j) f_in[j] = freq[(f0 + j < n) ? f0 + j : n - 1];
        }
        if (phase) {
#pragma unroll
            for (int j = 0; j < kGateT; ++j) ph_in[j] = phase[(f0 + j < n) ? f0 + j : n - 1];
        }
        if (duty) {
#pragma unroll
            for (int j = 0; j < kGateT; ++j) d_in[j] = duty[(f0 + j < n) ? f0 + j : n - 1];
        }
#pragma unroll
        for (int j = 0; j < kGateT; ++j) {
            loc[j] = run;                                             // exclusive within the chunk
            const double f = freq ? ((f0 + j < n) ? (double)f_in[j] : 0.0) : freq_scalar;
            run = run + ((f0 + j < n) ? f / sr : 0.0);
        }
        double tile_total;
        const double off = block_excl_sum(run, lds, tile_total);
        const double chunk_base = carry_sum + off;
        carry_sum = carry_sum + tile_total;
#pragma unroll
        for (int j = 0; j < kGateT; ++j) {
            if (f0 + j >= n) break;
            const double b = pgx::pgx_mod1(phase0 + (chunk_base + loc[j]));
            const double ph = pgx::pgx_mod1(b + (phase ? (double)ph_in[j] : phase_scalar));
            double d = duty ? (double)d_in[j] : duty_scalar;
            d = d < 0.0 ? 0.0 : (d > 1.0 ? 1.0 : d);
            out[f0 + j] = (ph < d) ? 1.0f : 0.0f;
        }
    }
    if (tid == 0) state[0] = pgx::pgx_mod1(phase0 + carry_sum);       // mod(phase + sum(dt), 1)
}

// ================================================================================================
// SVFilterPE (svfilter_pe.py:41-205, 404-500): trapezoidal state variable filter,
//   out = c0*x + c1*s0 + c2*s1;   s' = B*x + A*s   with a full (possibly per-sample) 2x2 A.
// One workgroup per channel chain, time-varying 2x2 affine scan like the varying biquad.
// ================================================================================================
constexpr int kSvT = 4;
constexpr int kSvTile = kBlock * kSvT;

struct SvCoef {
    double a00, a01, a10, a11, b0, b1, c0, c1, c2;
};

// svfilter_pe.py:120-205, per-sample scalar arithmetic in the reference's order.
__device__ __forceinline__ SvCoef svf_coef(int mode, double freq, double q, double a_lin, double sr) {
    double f_norm = pgx::pgx_div_fast(freq, sr);        // (divisions: see rbj)
    if (f_norm < 1e-6) f_norm = 1e-6;
    if (f_norm > 0.5) f_norm = 0.5;
    double res;
    if (mode == 4) {                                   // peaking ("bell")
        double qc = q < 0.01 ? 0.01 : (q > 100.0 ? 100.0 : q);
        const double k_bell = pgx::pgx_div_fast(1.0, qc * a_lin);
        res = 1.0 - 0.5 * k_bell;
    } else {
        double qc = q < 0.01 ? 0.01 : (q > 100.0 ? 100.0 : q);
        res = 1.0 - pgx::pgx_div_fast(0.5, qc);
    }
    if (res < 0.0) res = 0.0;
    if (res > 0.999) res = 0.999;
    const double k = 2.0 - 2.0 * res;
    double sn, cs;
    pgx::pgx_sincos_bounded(kPi * f_norm, sn, cs);     // <= pi/2
    double g = pgx::pgx_div_fast(sn, cs);              // tan(pi * f_norm)
    double shelf_a = 1.0;
    if (mode == 5) shelf_a = 1.0 / sqrt(a_lin);
    else if (mode == 6) shelf_a = sqrt(a_lin);
    g = g * shelf_a;
    const double a1 = pgx::pgx_div_fast(1.0, 1.0 + g * (g + k));
    const double a2 = g * a1;
    const double a3 = g * a2;
    SvCoef c;
    c.a00 = 2.0 * a1 - 1.0;
    c.a01 = -2.0 * a2;
    c.a10 = 2.0 * a2;
    c.a11 = 1.0 - 2.0 * a3;
    c.b0 = 2.0 * a2;
    c.b1 = 2.0 * a3;
    double m0, m1, m2;
    switch (mode) {
    case 0: m0 = 0.0; m1 = 0.0; m2 = 1.0; break;
    case 1: m0 = 1.0; m1 = -k; m2 = -1.0; break;
    case 2: m0 = 0.0; m1 = 1.0; m2 = 0.0; break;
    case 3: m0 = 1.0; m1 = -k; m2 = 0.0; break;
    case 4: m0 = 1.0; m1 = k * (a_lin * a_lin - 1.0); m2 = 0.0; break;
    case 5: m0 = 1.0; m1 = k * (a_lin - 1.0); m2 = a_lin * a_lin - 1.0; break;
    default: {
        const double A2 = a_lin * a_lin;
        m0 = A2; m1 = k * (a_lin - A2); m2 = 1.0 - A2;
    } break;
    }
    c.c0 = m0 * 1.0 + m1 * a2 + m2 * a3;
    c.c1 = m1 * a1 + m2 * a2;
    c.c2 = -m1 * a2 + m2 * (1.0 - a3);
    return c;
}

// state[channel] = {s0, s1}; snapshot / agg as for the varying biquad.
template <int MODE>
__global__ void __launch_bounds__(kBlock)
k_svf(float *out, const float *in, int64_t n, int channels, double sr, const pgx_biquad_var_params *params,
      const float *freq, const float *qs, double a_lin, const double *coef, double *state, double *snapshot,
      double *agg, int seg_tiles, int nseg) {
    __shared__ BvShared sh;
    const int tid = threadIdx.x;
    const int seg = blockIdx.x, ch = blockIdx.y;
    const pgx_biquad_var_params p = params[0];
    double *st = state + ch * 2;
    const double *init = (MODE == 1) ? snapshot + ch * 2 : st;
    double *ag = agg + ((int64_t)ch * nseg) * 6;
    V2 carry{init[0], init[1]};
    if (MODE == 1) carry = scan2_fold(ag, seg, carry);
    if (MODE == 0 && seg == 0 && tid < 2) snapshot[ch * 2 + tid] = st[tid];
    M2 acc_m = m_identity();
    V2 acc_v{0.0, 0.0};
    V2 final_s{0.0, 0.0};
    bool have_final = false;
    const bool varying = (freq != nullptr) || (qs != nullptr);
    SvCoef cconst;
    if (coef) {                                          // host-evaluated constant coefficients
        cconst = SvCoef{coef[0], coef[1], coef[2], coef[3], coef[4], coef[5], coef[6], coef[7], coef[8]};
    } else {
        cconst = svf_coef(p.mode, p.freq, p.q, a_lin, sr);
    }

    const int64_t tile0 = (int64_t)seg * seg_tiles;
    for (int t = 0; t < seg_tiles; ++t) {
        const int64_t base = (tile0 + t) * kSvTile;
        if (base >= n) break;
        const int64_t f0 = base + (int64_t)tid * kSvT;
        SvCoef cf[kSvT];
        double xs[kSvT];
        M2 cm = m_identity();
        V2 cv{0.0, 0.0};
        // samples and control values first, unconditionally (see k_biquad_varying)
        float x_in[kSvT], f_in[kSvT], q_in[kSvT];
#pragma unroll
        for (int j = 0; j < kSvT; ++j) {
            const int64_t at = (f0 + j < n) ? f0 + j : n - 1;
            x_in[j] = in[at * channels + ch];
        }
        if (varying && freq) {
#pragma unroll
            for (int j = 0; j < kSvT; ++j) f_in[j] = freq[(f0 + j < n) ? f0 + j : n - 1];
        }
        if (varying && qs) {
#pragma unroll
            for (int j = 0; j < kSvT; ++j) q_in[j] = qs[(f0 + j < n) ? f0 + j : n - 1];
        }
#pragma unroll
        for (int j = 0; j < kSvT; ++j) {
            const bool live = (f0 + j < n);
            xs[j] = live ? (double)x_in[j] : 0.0;
            if (varying) {
                double f = p.freq, q = p.q;
                if (freq) f = live ? (double)f_in[j] : 1000.0;
                if (qs) q = live ? (double)q_in[j] : 1.0;
                cf[j] = svf_coef(p.mode, f, q, a_lin, sr);
            } else {
                cf[j] = cconst;
            }
            if (live) {                                      // compose s' = A s + B x
                const M2 A{cf[j].a00, cf[j].a01, cf[j].a10, cf[j].a11};
                const V2 bx{cf[j].b0 * xs[j], cf[j].b1 * xs[j]};
                cv = vadd(mv(A, cv), bx);
                cm = mm(A, cm);
            }
        }
        V2 s = scan2_tile<MODE>(sh, cm, cv, carry, acc_m, acc_v);
        if (MODE == 0) continue;

        float yf[kSvT];
#pragma unroll
        for (int j = 0; j < kSvT; ++j) {
            const double xn = xs[j];
            const double y = cf[j].c0 * xn + cf[j].c1 * s.x + cf[j].c2 * s.y;      // svfilter_pe.py:58,86
            yf[j] = (float)y;
            if (f0 + j < n) {
                const double n0 = cf[j].b0 * xn + cf[j].a00 * s.x + cf[j].a01 * s.y;
                const double n1 = cf[j].b1 * xn + cf[j].a10 * s.x + cf[j].a11 * s.y;
                s.x = n0;
                s.y = n1;
            }
            if (f0 + j == n - 1) {
                final_s = s;
                have_final = true;
            }
        }
        store_frames<kSvT>(out, f0, n, channels, ch, yf);
    }
    if (MODE == 0) {
        if (tid == 0) {
            double *a = ag + (int64_t)seg * 6;
            a[0] = acc_m.a; a[1] = acc_m.b; a[2] = acc_m.c; a[3] = acc_m.d; a[4] = acc_v.x; a[5] = acc_v.y;
        }
        return;
    }
    if (have_final) {
        st[0] = final_s.x;
        st[1] = final_s.y;
    }
}

// ================================================================================================
// EnvelopePE (envelope_pe.py:128-271)
// ================================================================================================
// Detector: |x| (peak) or the block-local centred running RMS of scipy.ndimage.uniform_filter1d(x^2,
// size=window, mode='nearest') (envelope_pe.py:208-225).  Output float64 (frames, channels).
// RMS sums each window as head + whole 64-frame blocks + tail (k_env_blocks holds the block sums of squares),
// with the 'nearest' edge samples weighted by how often the clamped window repeats them.
constexpr int kEnvBlock = 64;

__global__ void __launch_bounds__(kBlock)
k_env_blocks(double *blocks, const float *in, int64_t n, int channels) {
    const int lane = threadIdx.x & 63;
    const int64_t nblk = (n + kEnvBlock - 1) / kEnvBlock;
    const int64_t item = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);          // one wave per (block, channel)
    if (item >= nblk * channels) return;
    const int64_t b = item / channels;
    const int c = (int)(item - b * channels);
    const int64_t f = b * kEnvBlock + lane;
    double v = 0.0;
    if (f < n) {
        v = (double)in[f * channels + c];
        v = v * v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = v + __shfl_down(v, d, 64);
    if (lane == 0) blocks[item] = v;
}

__global__ void __launch_bounds__(kBlock)
k_env_detect(double *det, const float *in, const double *blocks, int64_t n, int channels, int rms_window,
             int64_t period) {
    const int64_t total = n * channels;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    auto sq = [&](int64_t f, int c) {
        const double v = (double)in[f * channels + c];
        return v * v;
    };
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t i = e / channels;
        const int c = (int)(e - i * channels);
        if (rms_window <= 0) {
            det[e] = fabs((double)in[e]);
            continue;
        }
        // window [i - size/2, i - size/2 + size) with indices clamped to the block ('nearest').  The block is the
        // caller's: when a look-ahead window renders several of the caller's blocks at once (`period` frames each)
        // the detector restarts at every one of their edges, as the reference's per-block filter does
        int64_t p0 = 0, p1 = n;
        if (period > 0) {
            p0 = (i / period) * period;
            p1 = p0 + period < n ? p0 + period : n;
        }
        int64_t lo = i - rms_window / 2, hi = lo + rms_window;
        double acc = 0.0;
        if (lo < p0) {
            acc = acc + (double)(p0 - lo) * sq(p0, c);
            lo = p0;
        }
        if (hi > p1) {
            acc = acc + (double)(hi - p1) * sq(p1 - 1, c);
            hi = p1;
        }
        const int64_t b0 = (lo + kEnvBlock - 1) / kEnvBlock, b1 = hi / kEnvBlock;        // whole blocks [b0, b1)
        if (b0 < b1) {
            for (int64_t f = lo; f < b0 * kEnvBlock; ++f) acc = acc + sq(f, c);
            for (int64_t b = b0; b < b1; ++b) acc = acc + blocks[b * channels + c];
            for (int64_t f = b1 * kEnvBlock; f < hi; ++f) acc = acc + sq(f, c);
        } else {
            for (int64_t f = lo; f < hi; ++f) acc = acc + sq(f, c);
        }
        det[e] = sqrt(acc / (double)rms_window);
    }
}

constexpr int kEnvT = 8;
constexpr int kEnvTile = kBlock * kEnvT;

// attack == release (envelope_pe.py:167-180): one-pole lfilter  y = z + c*x;  z = (1-c)*y.
__global__ void __launch_bounds__(kBlock)
k_env_onepole(float *out, const double *det, int64_t n, int channels, double coeff, double *state) {
    __shared__ double lds[kWaves];
    const int tid = threadIdx.x, lane = tid & 63;
    const int ch = blockIdx.x;
    const double lam = 1.0 - coeff;
    double lamp[6], lam_wave, lam_lane = 1.0;
    {
        double l = lam;
#pragma unroll
        for (int s = 1; s < kEnvT; s <<= 1) l = l * l;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            lamp[k] = l;
            if (lane & (1 << k)) lam_lane = lam_lane * l;
            l = l * l;
        }
        lam_wave = l;
    }
    double carry = state[ch];
    double final_y = 0.0;
    bool have_final = false;
    for (int64_t base = 0; base < n; base += kEnvTile) {
        const int64_t f0 = base + (int64_t)tid * kEnvT;
        double xs[kEnvT];
        double e = 0.0;
#pragma unroll
        for (int j = 0; j < kEnvT; ++j) {
            xs[j] = (f0 + j < n) ? det[(f0 + j) * channels + ch] : 0.0;
            // dead samples must act as the identity map, so the zero-state response only folds live ones
            if (f0 + j < n) e = lam * e + coeff * xs[j];
            else e = lam * e;
        }
        double y = block_scan_scalar_affine(e, lamp, lam_wave, lam_lane, lds, carry);
        float yf[kEnvT];
#pragma unroll
        for (int j = 0; j < kEnvT; ++j) {
            const double z = lam * y;
            const double yn = z + coeff * xs[j];
            yf[j] = (float)yn;
            if (f0 + j < n) y = yn;
            if (f0 + j == n - 1) {
                final_y = yn;
                have_final = true;
            }
        }
        store_frames<kEnvT>(out, f0, n, channels, ch, yf);
    }
    if (have_final) state[ch] = final_y;
}


// attack != release (envelope_pe.py:259-271), time-parallel.  One sample's update
//     e' = e + c(e) * (t - e),   c = attack if t > e else release
// is a continuous, increasing, two-piece linear map of e, so a thread's T samples compose to a piecewise linear
// map whose piece around a given entry level is found by stepping the samples literally from that level.  A
// window of NW*64*T samples is solved by Newton's method on that piecewise linear system: every thread steps its
// samples from its current entry level (reference arithmetic), recording the affine piece (A, B) it passed
// through; a workgroup scan of the pieces gives every thread a new entry level; repeat until no entry level
// moves by more than 1e-13 of itself.  Thread k's entry is final after at most k+1 rounds (it only depends on
// threads before it), so the loop ends; in practice it takes 2..6 rounds (semismooth Newton).  The samples written
// are literal steps from entry levels within 1e-13 of the converged ones: against the sequential reference they
// differ by that much (float32 output resolves 6e-8).  `det` == nullptr: peak detection fused (|in|).
constexpr double kEnvSettled = 1e-13;
template <int NW, int T, bool PEAK>
__global__ void __launch_bounds__(NW * 64)
k_env_newton(float *out, const float *in, const double *det, int64_t n, int channels, double attack_coeff,
             double release_coeff, double *state, const int *run_flag) {
    if (run_flag != nullptr && *run_flag == 0) return;       // armed only when the all-windows form gave up
    __shared__ double s_a[2][NW], s_b[2][NW], s_pa[T + 1], s_pr[T + 1];
    __shared__ int s_moved[2][NW];
    constexpr int kWindow = NW * 64 * T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = blockIdx.x;
    double e_in = state[ch];
    if (tid == 0) {                                   // (1-c)^k: a thread's slope is a product of these
        double pa = 1.0, pr = 1.0;
        for (int k = 0; k <= T; ++k) {
            s_pa[k] = pa;
            s_pr[k] = pr;
            pa = pa * (1.0 - attack_coeff);
            pr = pr * (1.0 - release_coeff);
        }
    }
    __syncthreads();

    // HBM is touched with lane-contiguous accesses (element i of thread tid is frame i*NW*64 + tid of the window);
    // the rounds want T consecutive frames per thread.  Both directions go through LDS, a chunk of T frames padded
    // to T+1 words so that neither access pattern conflicts.  (Thread-contiguous global accesses cost 8x the L1->L2
    // requests here: 2 us per window, as much as the rounds.)
    using Raw = typename std::conditional<PEAK, float, double>::type;
    constexpr int kThreads = NW * 64;
    __shared__ Raw s_x[kWindow + kWindow / T];
    __shared__ float s_y[kWindow + kWindow / T];
    Raw raw[T];
    auto fetch = [&](int64_t wbase) {                 // branch-free (clamped) so that the loads pipeline
#pragma unroll
        for (int i = 0; i < T; ++i) {
            int64_t f = wbase + i * kThreads + tid;
            f = f < n ? f : n - 1;
            if constexpr (PEAK) raw[i] = in[f * channels + ch];
            else raw[i] = det[f * channels + ch];
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int k = i * kThreads + tid;
            s_x[k + k / T] = raw[i];
        }
    };
    double t[T];
    auto take = [&]() {
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const Raw v = s_x[tid * (T + 1) + j];
            t[j] = PEAK ? fabs((double)v) : (double)v;
        }
    };
    fetch(0);
    stage();
    __syncthreads();
    take();
    for (int64_t base = 0; base < n; base += kWindow) {
        const int64_t f0 = base + (int64_t)tid * T;
        const int live = (n - f0 >= T) ? T : (n - f0 > 0 ? (int)(n - f0) : 0);
        const bool has_next = base + kWindow < n;
        const bool full = base + kWindow <= n;
        if (has_next) fetch(base + kWindow);                               // next window, in flight

        // Two barriers per round (pieces, then the "an entry level moved" flags); LDS is double-buffered by round
        // parity so that no third one is needed.
        double entry = e_in, exit_level = e_in;
        double y[T];
        for (int round = 0; round <= NW * 64 + 1; ++round) {
            double e = entry;
            int attacks = 0;
            if (full) {                                  // every thread owns T live samples: no per-sample guard
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    const bool attack = t[j] > e;
                    e = e + (attack ? attack_coeff : release_coeff) * (t[j] - e);
                    attacks += attack ? 1 : 0;
                    y[j] = e;
                }
            } else {
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    const bool on = j < live;
                    const bool attack = t[j] > e;
                    const double stepped = e + (attack ? attack_coeff : release_coeff) * (t[j] - e);
                    attacks += (on && attack) ? 1 : 0;
                    e = on ? stepped : e;
                    y[j] = e;
                }
            }
            // the affine piece this thread passed through: slope from the regime counts, offset from the exit
            double a = s_pa[attacks] * s_pr[live - attacks];
            double b = __builtin_fma(-a, entry, e);
            // inclusive scan of the pieces over the wave (lanes without a source see the identity)
#define PGX_ENV_STEP(CTRL, MASK)                                            \
            {                                                               \
                const double ao = dpp_f64_keep<CTRL, MASK>(1.0, a);         \
                const double bo = dpp_f64_keep<CTRL, MASK>(0.0, b);         \
                b = __builtin_fma(a, bo, b);                                \
                a = a * ao;                                                 \
            }
            PGX_ENV_STEP(0x111, 0xf) PGX_ENV_STEP(0x112, 0xf) PGX_ENV_STEP(0x114, 0xf) PGX_ENV_STEP(0x118, 0xf)
            PGX_ENV_STEP(0x142, 0xa) PGX_ENV_STEP(0x143, 0xc)
#undef PGX_ENV_STEP
            const int buf = round & 1;
            if (lane == 63) {
                s_a[buf][wave] = a;
                s_b[buf][wave] = b;
            }
            __syncthreads();
            double cw = e_in, cn = e_in;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const double wa = s_a[buf][w], wb = s_b[buf][w];
                if (w < wave) cw = __builtin_fma(wa, cw, wb);
                cn = __builtin_fma(wa, cn, wb);
            }
            exit_level = cn;
            const double aex = dpp_f64_keep<0x138, 0xf>(1.0, a);            // wave_shr:1 -> exclusive
            const double bex = dpp_f64_keep<0x138, 0xf>(0.0, b);
            const double fresh = __builtin_fma(aex, cw, bex);
            // settled when no entry level moves by more than rounding noise (the offsets above carry ~1e-16)
            const bool moved = live > 0 && fabs(fresh - entry) > kEnvSettled * (fabs(fresh) + fabs(entry));
            entry = fresh;
            const bool wave_moved = __ballot(moved) != 0ull;
            if (lane == 0) s_moved[buf][wave] = wave_moved ? 1 : 0;
            __syncthreads();
            int any = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) any |= s_moved[buf][w];
            if (!any) break;
        }
#pragma unroll
        for (int j = 0; j < T; ++j) s_y[tid * (T + 1) + j] = (float)y[j];
        if (has_next) stage();
        __syncthreads();                               // (the next writes to s_x / s_y come after the next rounds' barriers)
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int k = i * kThreads + tid;
            if (base + k < n) out[(base + k) * channels + ch] = s_y[k + k / T];
        }
        if (has_next) take();
        e_in = exit_level;                           // composition of all pieces (dead samples are the identity)
    }
    if (tid == 0) state[ch] = e_in;
}

// The same follower over a LONG block (look-ahead windows hand it millions of frames): all 8192-sample windows
// at once.  A window's exit level is an affine function of its entry level on the piece the trajectory runs
// through (A = product of the per-sample slopes, tiny after thousands of samples), so the entry levels of all
// windows solve a piecewise linear system of their own, again by Newton rounds -- one launch per round:
//   round 0   every window is solved (the inner rounds above) from the carried level and publishes its piece;
//   round r   a window folds the pieces of the windows before it (published in round r-1) onto the carried
//             level; if that entry is the one its samples were rendered from (1e-13) it republishes its piece and
//             is done, otherwise it renders again from the new entry and raises the round's "moved" flag.
// No window moved in a round = every window was rendered from the entry the others imply: converged; later
// launches return at once.  Window w is final after at most w+1 rounds, audio settles in 2..4 (a window forgets
// its entry by orders of magnitude).  If kEnvMwRounds are not enough the finishing kernel arms the sequential
// kernel above, which then renders the block from the carried state: the result never depends on convergence.
constexpr int kEnvMwRounds = 8;
constexpr int kEnvMwNW = 8, kEnvMwT = 16, kEnvMwWindow = kEnvMwNW * 64 * kEnvMwT;      // 8192
struct EnvMwCtl {
    int moved[kEnvMwRounds + 1];
    int fallback;
};
template <bool PEAK>
__global__ void __launch_bounds__(kEnvMwNW * 64)
k_env_newton_mw(float *out, const float *in, const double *det, int64_t n, int channels, double attack_coeff,
                double release_coeff, const double *state, int round, int nwin, double *pa_buf, double *pb_buf,
                double *guess, EnvMwCtl *ctl) {
    constexpr int NW = kEnvMwNW, T = kEnvMwT, kWindow = kEnvMwWindow, kThreads = NW * 64;
    __shared__ double s_a[2][NW], s_b[2][NW], s_pa[T + 1], s_pr[T + 1], s_entry;
    __shared__ int s_moved[2][NW];
    using Raw = typename std::conditional<PEAK, float, double>::type;
    __shared__ Raw s_x[kWindow + kWindow / T];
    __shared__ float s_y[kWindow + kWindow / T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = blockIdx.x, ch = blockIdx.y;
    if (round > 0 && ctl->moved[round - 1] == 0) return;         // converged in an earlier round
    const int64_t base = (int64_t)w * kWindow;
    const int cur = round & 1, prev = cur ^ 1;
    double *pa_cur = pa_buf + ((int64_t)cur * channels + ch) * nwin, *pb_cur = pb_buf + ((int64_t)cur * channels + ch) * nwin;
    const double *pa_prev = pa_buf + ((int64_t)prev * channels + ch) * nwin;
    const double *pb_prev = pb_buf + ((int64_t)prev * channels + ch) * nwin;

    // request the window's frames first: their latency covers the fold below
    Raw raw[T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
        int64_t f = base + i * kThreads + tid;
        f = f < n ? f : n - 1;
        if constexpr (PEAK) raw[i] = in[f * channels + ch];
        else raw[i] = det[f * channels + ch];
    }
    if (tid == 0) {
        double pa = 1.0, pr = 1.0;
        for (int k = 0; k <= T; ++k) {
            s_pa[k] = pa;
            s_pr[k] = pr;
            pa = pa * (1.0 - attack_coeff);
            pr = pr * (1.0 - release_coeff);
        }
    }
    // entry level: the carried level pushed through the pieces of windows 0..w-1 (wave 0: lanes compose runs of
    // pieces, then a scan over the lanes)
    if (wave == 0) {
        double a = 1.0, b = 0.0;
        if (round > 0) {
            const int per = (w + 63) / 64;
            const int v0 = lane * per, v1 = (v0 + per < w) ? v0 + per : w;
            for (int v = v0; v < v1; ++v) {
                const double av = pa_prev[v], bv = pb_prev[v];
                b = __builtin_fma(av, b, bv);
                a = a * av;
            }
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const double ao = __shfl_up(a, d, 64), bo = __shfl_up(b, d, 64);
                if (lane >= d) {
                    b = __builtin_fma(a, bo, b);
                    a = a * ao;
                }
            }
        }
        if (lane == 63) s_entry = __builtin_fma(a, state[ch], b);
    }
    __syncthreads();
    const double e_in = s_entry;
    if (round > 0) {
        const double g = guess[(int64_t)ch * nwin + w];
        if (fabs(e_in - g) <= kEnvSettled * (fabs(e_in) + fabs(g))) {      // rendered from this entry already
            if (tid == 0) {
                pa_cur[w] = pa_prev[w];
                pb_cur[w] = pb_prev[w];
            }
            return;
        }
    }
    if (tid == 0) {
        ctl->moved[round] = 1;
        guess[(int64_t)ch * nwin + w] = e_in;
    }
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int k = i * kThreads + tid;
        s_x[k + k / T] = raw[i];
    }
    __syncthreads();
    double t[T];
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const Raw v = s_x[tid * (T + 1) + j];
        t[j] = PEAK ? fabs((double)v) : (double)v;
    }
    const int64_t f0 = base + (int64_t)tid * T;
    const int live = (n - f0 >= T) ? T : (n - f0 > 0 ? (int)(n - f0) : 0);
    double entry = e_in, a_tot = 1.0, b_tot = 0.0;
    double y[T];
    for (int rnd = 0; rnd <= NW * 64 + 1; ++rnd) {
        double e = entry;
        int attacks = 0;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const bool on = j < live;
            const bool attack = t[j] > e;
            const double stepped = e + (attack ? attack_coeff : release_coeff) * (t[j] - e);
            attacks += (on && attack) ? 1 : 0;
            e = on ? stepped : e;
            y[j] = e;
        }
        double a = s_pa[attacks] * s_pr[live - attacks];
        double b = __builtin_fma(-a, entry, e);
#define PGX_ENV_STEP(CTRL, MASK)                                            \
        {                                                                   \
            const double ao = dpp_f64_keep<CTRL, MASK>(1.0, a);             \
            const double bo = dpp_f64_keep<CTRL, MASK>(0.0, b);             \
            b = __builtin_fma(a, bo, b);                                    \
            a = a * ao;                                                     \
        }
        PGX_ENV_STEP(0x111, 0xf) PGX_ENV_STEP(0x112, 0xf) PGX_ENV_STEP(0x114, 0xf) PGX_ENV_STEP(0x118, 0xf)
        PGX_ENV_STEP(0x142, 0xa) PGX_ENV_STEP(0x143, 0xc)
#undef PGX_ENV_STEP
        const int buf = rnd & 1;
        if (lane == 63) {
            s_a[buf][wave] = a;
            s_b[buf][wave] = b;
        }
        __syncthreads();
        double cw = e_in, ta = 1.0, tb = 0.0;
#pragma unroll
        for (int v = 0; v < NW; ++v) {
            const double wa = s_a[buf][v], wb = s_b[buf][v];
            if (v < wave) cw = __builtin_fma(wa, cw, wb);
            tb = __builtin_fma(wa, tb, wb);                       // the window's piece: exit = ta * entry + tb
            ta = ta * wa;
        }
        a_tot = ta;
        b_tot = tb;
        const double aex = dpp_f64_keep<0x138, 0xf>(1.0, a);
        const double bex = dpp_f64_keep<0x138, 0xf>(0.0, b);
        const double fresh = __builtin_fma(aex, cw, bex);
        const bool moved = live > 0 && fabs(fresh - entry) > kEnvSettled * (fabs(fresh) + fabs(entry));
        entry = fresh;
        const bool wave_moved = __ballot(moved) != 0ull;
        if (lane == 0) s_moved[buf][wave] = wave_moved ? 1 : 0;
        __syncthreads();
        int any = 0;
#pragma unroll
        for (int v = 0; v < NW; ++v) any |= s_moved[buf][v];
        if (!any) break;
    }
#pragma unroll
    for (int j = 0; j < T; ++j) s_y[tid * (T + 1) + j] = (float)y[j];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int k = i * kThreads + tid;
        if (base + k < n) out[(base + k) * channels + ch] = s_y[k + k / T];
    }
    if (tid == 0) {
        pa_cur[w] = a_tot;
        pb_cur[w] = b_tot;
    }
}

// After the last round: converged -> the new carried level is the carried level pushed through every piece;
// otherwise arm the sequential kernel.
__global__ void __launch_bounds__(64)
k_env_mw_finish(double *state, int channels, int nwin, const double *pa_buf, const double *pb_buf, EnvMwCtl *ctl,
                int rounds) {
    const int lane = threadIdx.x;
    int last = 0;                                                 // the last round in which a window moved
    for (int r = 0; r < rounds; ++r)
        if (ctl->moved[r]) last = r;
    const bool converged = last < rounds - 1;                     // a later round ran and found nothing to move
    if (!converged) {
        if (lane == 0 && blockIdx.x == 0) ctl->fallback = 1;
        return;
    }
    // pieces of the converged configuration: published by the last round that ran to its end, i.e. `last` (a round
    // in which nothing moved republishes the same pieces); round last + 1, if launched, returned at once
    const int ch = blockIdx.x;
    const int cur = (last + 1) & 1;                               // that later round republished every piece
    const double *pa = pa_buf + ((int64_t)cur * channels + ch) * nwin, *pb = pb_buf + ((int64_t)cur * channels + ch) * nwin;
    double a = 1.0, b = 0.0;
    const int per = (nwin + 63) / 64;
    const int v0 = lane * per, v1 = (v0 + per < nwin) ? v0 + per : nwin;
    for (int v = v0; v < v1; ++v) {
        b = __builtin_fma(pa[v], b, pb[v]);
        a = a * pa[v];
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double ao = __shfl_up(a, d, 64), bo = __shfl_up(b, d, 64);
        if (lane >= d) {
            b = __builtin_fma(a, bo, b);
            a = a * ao;
        }
    }
    if (lane == 63) state[ch] = __builtin_fma(a, state[ch], b);
}

// ================================================================================================
// TransformPE: chains of named element-wise float64 operations (pygmu2_amd/transforms.py)
// ================================================================================================
__global__ void __launch_bounds__(kBlock)
k_transform(float *out, const float *in, int64_t n_elems, const pgx_transform_op *ops, int nops) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n_elems; e += stride) {
        double v = (double)in[e];
        for (int k = 0; k < nops; ++k) {
            const pgx_transform_op op = ops[k];
            switch (op.code) {
            case 0: v = op.p1 + op.p0 * v; break;                              // affine: offset + scale*x
            case 1: v = v < op.p0 ? op.p0 : (v > op.p1 ? op.p1 : v); break;    // clip (NaN passes through)
            case 2: v = sqrt(v); break;                                        // x ** 0.5 (numpy -> sqrt)
            case 3: v = v * v; break;                                          // x ** 2 (numpy -> square)
            case 4: v = fabs(v); break;
            case 5: v = tanh(v); break;
            case 6: v = 1.0 - v; break;
            default: break;
            }
        }
        out[e] = (float)v;
    }
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

size_t pgx_biquad_workspace_bytes(int batch, int64_t n, int channels, int64_t settle_frames) {
    if (batch <= 0 || n <= 0 || channels <= 0) return 0;
    // a caller that passes settle_frames > 0 also passes tables; without them the exact pair may run
    if (settle_frames > 0 && biquad_settled_plan(batch, n, channels, settle_frames, true).ok) return 0;
    BqPlan p = biquad_plan(batch, n, channels);
    if (p.nseg <= 1) return 0;
    size_t chains = (size_t)batch * channels;
    return (chains * 2 + chains * (size_t)p.nseg * 2) * sizeof(double);
}

size_t pgx_biquad_table_doubles(void) { return kBqTableDoubles; }

int pgx_biquad_tables(double *tables, const double *coef, int batch) {
    PGX_REQUIRE_INIT();
    if (batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(tables && coef, "pgx_biquad_tables: bad argument");
    hipLaunchKernelGGL(k_biquad_tables, dim3(batch), dim3(64), 0, pgx::stream(), tables, coef);
    PGX_LAUNCH_CHECK("k_biquad_tables");
    return PGX_OK;
}

int pgx_biquad_const(float *out, int64_t out_stride, const float *in, int64_t in_stride, int batch, int64_t n,
                     int channels, const double *coef, const double *tables, int64_t settle_frames, double *state,
                     void *workspace) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && coef && state && channels >= 1, "pgx_biquad_const: bad argument");
    PGX_CHECK_ARG(batch == 1 || (out_stride >= n * channels && in_stride >= n * channels),
                  "pgx_biquad_const: instance stride too small");
    PGX_CHECK_ARG((int64_t)batch * channels <= 65535, "pgx_biquad_const: too many chains");
    int chains = batch * channels;
    const BqSettledPlan sp = biquad_settled_plan(batch, n, channels, settle_frames, tables != nullptr);
    if (sp.ok) {
        const dim3 grid(sp.groups == 1 ? 1 : (sp.groups + 7) / 8 * 8, chains);
        if (channels == 1)
            hipLaunchKernelGGL((k_biquad_settled<true, true>), grid, dim3(kSbBlock), 0, pgx::stream(), out,
                               out_stride, in, in_stride, n, channels, coef, tables, state, sp.seg, sp.head,
                               sp.tail, sp.warm, sp.groups);
        else
            hipLaunchKernelGGL((k_biquad_settled<false, false>), grid, dim3(kSbBlock), 0, pgx::stream(), out,
                               out_stride, in, in_stride, n, channels, coef, tables, state, sp.seg, sp.head,
                               sp.tail, sp.warm, sp.groups);
        PGX_LAUNCH_CHECK("k_biquad_settled");
        return PGX_OK;
    }
    BqPlan p = biquad_plan(batch, n, channels);
    dim3 grid(p.nseg, chains);
    if (p.nseg > 1) {
        PGX_CHECK_ARG(workspace != nullptr, "pgx_biquad_const: workspace required for this size");
        double *snap = (double *)workspace;
        double *agg = snap + (size_t)chains * 2;
        hipLaunchKernelGGL(k_biquad_const<0>, grid, dim3(kBlock), 0, pgx::stream(), out, out_stride, in,
                           in_stride, n, channels, coef, state, (const double *)snap, agg, p.seg_tiles, p.nseg);
        PGX_LAUNCH_CHECK("k_biquad_const<reduce>");
        hipLaunchKernelGGL(k_biquad_const<1>, grid, dim3(kBlock), 0, pgx::stream(), out, out_stride, in,
                           in_stride, n, channels, coef, state, (const double *)snap, agg, p.seg_tiles, p.nseg);
        PGX_LAUNCH_CHECK("k_biquad_const<apply>");
    } else {
        hipLaunchKernelGGL(k_biquad_const<1>, grid, dim3(kBlock), 0, pgx::stream(), out, out_stride, in,
                           in_stride, n, channels, coef, state, (const double *)nullptr, (double *)nullptr,
                           p.seg_tiles, p.nseg);
        PGX_LAUNCH_CHECK("k_biquad_const");
    }
    return PGX_OK;
}

// The sine-source variant: 4-wave workgroups, three per CU (PGX_SB_SINE_BLOCK=512: the 8-wave form, one per CU --
// experiments)
static int sine_block() {
    static const int block = getenv("PGX_SB_SINE_BLOCK") ? atoi(getenv("PGX_SB_SINE_BLOCK")) : 256;
    return block == 512 ? 512 : 256;
}
static BqSettledPlan biquad_sine_plan(int64_t n, int64_t settle_frames) {
    static const int want = getenv("PGX_SB_SINE_WGS") ? atoi(getenv("PGX_SB_SINE_WGS")) : (sine_block() == 256 ? 1024 : 512);
    return biquad_settled_plan(1, n, 1, settle_frames, true, want);
}

int pgx_biquad_sine_supported(int64_t n, int64_t settle_frames) {
    return (n > 0 && biquad_sine_plan(n, settle_frames).ok) ? 1 : 0;
}

int pgx_biquad_sine(float *out, int64_t start, int64_t n, double sample_rate, double w, double amp, double phase0,
                    const double *coef, const double *tables, int64_t settle_frames, double *state,
                    double *state_backup) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && coef && tables && state && sample_rate > 0, "pgx_biquad_sine: bad argument");
    const int block = sine_block();
    const BqSettledPlan sp = biquad_sine_plan(n, settle_frames);
    PGX_CHECK_ARG(sp.ok, "pgx_biquad_sine: block too short for the single-launch plan (pgx_biquad_sine_supported)");
    // the largest phase of the render has to stay in the bounded sine's range
    const double far = fabs(phase0) + fabs(w) * ((double)(llabs(start) + n) / sample_rate);
    PGX_CHECK_ARG(far < pgx::kSinFastRange, "pgx_biquad_sine: phase beyond the fast sine range");
    // the tone's constants (five long-double sines and cosines) are made once per (w, sample rate): a stream of windows
    // asks for the same tone every time, and they were 1 us of every window opening's host time
    static SbSine cached;
    static bool have = false;
    if (!have || cached.w != w || cached.sr != sample_rate) {
        SbSine c;
        c.w = w;
        c.sr = sample_rate;
        c.inv_sr = 1.0 / sample_rate;
        const long double d = (long double)w / (long double)sample_rate;
        c.cos_d = (double)cosl(d);
        c.sin_d = (double)sinl(d);
        c.two_cos_d = (fabsl(sinl(d)) >= 1e-3L) ? (double)(2.0L * cosl(d)) : 0.0;
        // a tile's advance, block * 16 frames: the angle reduced in long double before the sine and cosine are taken
        const long double turn = 6.283185307179586476925286766559L;
        long double a = d * (long double)(sine_block() * kBqT);
        a -= turn * floorl(a / turn);
        c.tile_cos = (double)cosl(a);
        c.tile_sin = (double)sinl(a);
        cached = c;
        have = true;
    }
    SbSine sine = cached;
    sine.amp = amp;
    sine.phase0 = phase0;
    sine.start = start;
    sine.state_backup = state_backup;
    const dim3 grid(sp.groups == 1 ? 1 : (sp.groups + 7) / 8 * 8, 1);
    if (block == 256)
        hipLaunchKernelGGL((k_biquad_settled<true, true, true, 256>), grid, dim3(256), 0, pgx::stream(), out,
                           (int64_t)0, (const float *)nullptr, (int64_t)0, n, 1, coef, tables, state, sp.seg, sp.head,
                           sp.tail, sp.warm, sp.groups, sine);
    else
        hipLaunchKernelGGL((k_biquad_settled<true, true, true>), grid, dim3(kSbBlock), 0, pgx::stream(), out,
                           (int64_t)0, (const float *)nullptr, (int64_t)0, n, 1, coef, tables, state, sp.seg, sp.head,
                           sp.tail, sp.warm, sp.groups, sine);
    PGX_LAUNCH_CHECK("k_biquad_settled<sine>");
    return PGX_OK;
}

// A = 10^(gain_db/40) is a host-side Python float pow in the reference (biquad_pe.py:250); the
// binding passes it and its square root by value so the device never calls pow().
size_t pgx_scan2_workspace_bytes(int64_t n, int channels) {
    if (n <= 0 || channels <= 0) return 0;
    const Scan2Plan p = scan2_plan(n, kBvTile);
    if (p.nseg <= 1) return 0;
    return (size_t)channels * (4 + (size_t)p.nseg * 6) * sizeof(double);
}

int pgx_biquad_varying(float *out, const float *in, int64_t n, int channels, double sample_rate,
                       const pgx_biquad_var_params *params, const float *freq, const float *q,
                       double gain_a, double gain_sqrt_a, double *state, void *workspace) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && params && state && channels >= 1 && sample_rate > 0,
                  "pgx_biquad_varying: bad argument");
    const Scan2Plan p = scan2_plan(n, kBvTile);
    double *snap = (double *)workspace;
    double *agg = snap ? snap + (size_t)channels * 4 : nullptr;
    if (p.nseg <= 1 || workspace == nullptr) {
        hipLaunchKernelGGL(k_biquad_varying<2>, dim3(1, channels), dim3(kBlock), 0, pgx::stream(), out, in, n,
                           channels, sample_rate, params, freq, q, gain_a, gain_sqrt_a, state, snap, agg,
                           (int)pgx::ceil_div(n, kBvTile), 1);
        PGX_LAUNCH_CHECK("k_biquad_varying");
        return PGX_OK;
    }
    hipLaunchKernelGGL(k_biquad_varying<0>, dim3(p.nseg, channels), dim3(kBlock), 0, pgx::stream(), out, in, n,
                       channels, sample_rate, params, freq, q, gain_a, gain_sqrt_a, state, snap, agg, p.seg_tiles,
                       p.nseg);
    PGX_LAUNCH_CHECK("k_biquad_varying<reduce>");
    hipLaunchKernelGGL(k_biquad_varying<1>, dim3(p.nseg, channels), dim3(kBlock), 0, pgx::stream(), out, in, n,
                       channels, sample_rate, params, freq, q, gain_a, gain_sqrt_a, state, snap, agg, p.seg_tiles,
                       p.nseg);
    PGX_LAUNCH_CHECK("k_biquad_varying<apply>");
    return PGX_OK;
}

size_t pgx_blitsaw_workspace_bytes(int batch, int64_t n, int streams) {
    if (batch <= 0 || n <= 0) return 0;
    const SawPlan p = saw_plan(batch, n, streams != 0);
    return p.nseg > 1 ? (size_t)batch * (2 + (size_t)p.tiles * (kSawWideWaves + 1) + (size_t)p.nseg) * sizeof(double) : 0;
}

int pgx_blitsaw(float *out, int64_t out_stride, int batch, int64_t n, int channels, double sample_rate,
                const pgx_blitsaw_params *params, const float *freq, int64_t freq_stride, const float *amp,
                int64_t amp_stride, const float *m, int64_t m_stride, double *state, void *workspace,
                double *state_backup) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && params && state && channels >= 1 && sample_rate > 0, "pgx_blitsaw: bad argument");
    PGX_CHECK_ARG(batch == 1 || out_stride >= n * channels, "pgx_blitsaw: out_stride too small");
    // a few oscillators: 512-thread workgroups (half of the dependent tile steps, same bits);
    // a bank: 256-thread workgroups, the oscillators themselves fill the machine
    const bool wide = batch < 128 && n > kSawTile;
    const bool streams = freq || amp || m;
    const SawPlan plan = saw_plan(batch, n, streams);
#define PGX_SAW_LAUNCH(S, NWAVES, SEG, GRID)                                                                   \
    hipLaunchKernelGGL((k_blitsaw<S, NWAVES, SEG>), GRID, dim3(NWAVES * 64), 0, pgx::stream(), out, out_stride, \
                       n, channels, sample_rate, params, freq, freq_stride, amp, amp_stride, m, m_stride, state,  \
                       (double *)workspace, ws_stride, plan.tiles_per_seg, state_backup)
    const int64_t ws_stride = 2 + plan.tiles * (kSawWideWaves + 1) + (int64_t)plan.nseg;
    if (workspace && plan.nseg > 1) {        // long streams of a few oscillators: several workgroups each
        PGX_SAW_LAUNCH(false, kSawWideWaves, 1, dim3(batch, plan.nseg));
        PGX_LAUNCH_CHECK("k_blitsaw<reduce>");
        hipLaunchKernelGGL(k_blitsaw_chain<kSawWideWaves>, dim3(batch), dim3(64), 0, pgx::stream(), params,
                           (double *)workspace, ws_stride, plan.tiles);
        PGX_LAUNCH_CHECK("k_blitsaw_chain");
        PGX_SAW_LAUNCH(false, kSawWideWaves, 2, dim3(batch, plan.nseg));
    } else if (streams && wide) PGX_SAW_LAUNCH(true, kSawWideWaves, 0, dim3(batch));
    else if (streams) PGX_SAW_LAUNCH(true, kWaves, 0, dim3(batch));
    else if (wide) PGX_SAW_LAUNCH(false, kSawWideWaves, 0, dim3(batch));
    else PGX_SAW_LAUNCH(false, kWaves, 0, dim3(batch));
#undef PGX_SAW_LAUNCH
    PGX_LAUNCH_CHECK("k_blitsaw");
    return PGX_OK;
}

int pgx_blitsaw_biquad_bank(float *out, int64_t out_stride, int batch, int64_t n, double sample_rate,
                            const pgx_blitsaw_params *params, double *saw_state, const double *coef,
                            double *biquad_state) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && params && saw_state && coef && biquad_state && sample_rate > 0,
                  "pgx_blitsaw_biquad_bank: bad argument");
    PGX_CHECK_ARG(batch == 1 || out_stride >= n, "pgx_blitsaw_biquad_bank: out_stride too small");
    hipLaunchKernelGGL(k_blitsaw_biquad, dim3(batch), dim3(kWaves * 64), 0, pgx::stream(), out, out_stride, n,
                       sample_rate, params, saw_state, coef, biquad_state);
    PGX_LAUNCH_CHECK("k_blitsaw_biquad");
    return PGX_OK;
}

int pgx_blitsaw_biquad_wide(float *out, int64_t out_stride, int batch, int64_t n, const double *saw_tables,
                            double *saw_state, const double *coef, const double *biquad_tables,
                            double *biquad_state, const float *gain, int64_t gain_stride) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && saw_tables && saw_state && coef && biquad_tables && biquad_state,
                  "pgx_blitsaw_biquad_wide: bad argument");
    PGX_CHECK_ARG(batch == 1 || out_stride >= n, "pgx_blitsaw_biquad_wide: out_stride too small");
    PGX_CHECK_ARG(gain == nullptr || batch == 1 || gain_stride >= n, "pgx_blitsaw_biquad_wide: gain_stride too small");
    if (gain != nullptr)
        hipLaunchKernelGGL((k_blitsaw_biquad_wide<4, true>), dim3(batch), dim3(4 * 64), 0, pgx::stream(), out, out_stride,
                           n, saw_tables, saw_state, coef, biquad_tables, biquad_state, gain, gain_stride);
    else
        hipLaunchKernelGGL((k_blitsaw_biquad_wide<4, false>), dim3(batch), dim3(4 * 64), 0, pgx::stream(), out, out_stride,
                           n, saw_tables, saw_state, coef, biquad_tables, biquad_state, gain, gain_stride);
    PGX_LAUNCH_CHECK("k_blitsaw_biquad_wide");
    return PGX_OK;
}

// The same chain in concurrent time segments, for banks too small to fill the chip with one workgroup per voice (a
// rank's share of C5).  settle_frames: the filters' warm-up (biquad_pe.settle_frames: the largest of the bank; > 0).
// States travel from the *_in buffers to the *_out buffers (two different buffers each).
static int bbw_segment_tiles(int batch, int64_t n, int64_t settle_frames, int *warm_tiles_out) {
    constexpr int64_t tile = 4 * 64 * kSswT;
    const int64_t tiles = pgx::ceil_div(n, tile);
    const int64_t warm = pgx::ceil_div(settle_frames, tile);
    if (warm_tiles_out) *warm_tiles_out = (int)warm;
    static const int max_batch = getenv("PGX_BBW_MAX_BATCH") ? atoi(getenv("PGX_BBW_MAX_BATCH")) : 256;
    if (batch <= 0 || settle_frames <= 0 || batch > max_batch || tiles < 3) return (int)tiles;
    // enough workgroups for one per CU (a 4-wave workgroup alone on its CU renders a tile fastest), segments at
    // least twice their warm-up; PGX_BBW_WGS: experiments
    static const int wgs = getenv("PGX_BBW_WGS") ? atoi(getenv("PGX_BBW_WGS")) : pgx::kNumCU;
    int64_t want = pgx::ceil_div(wgs, batch);
    if (want < 1) want = 1;
    int64_t seg = pgx::ceil_div(tiles, want);
    if (seg < 2 * warm) seg = 2 * warm;
    if (seg > tiles) seg = tiles;
    return (int)seg;
}

int pgx_blitsaw_biquad_wide_segments(int batch, int64_t n, int64_t settle_frames) {
    if (batch <= 0 || n <= 0) return 1;
    constexpr int64_t tile = 4 * 64 * kSswT;
    const int seg = bbw_segment_tiles(batch, n, settle_frames, nullptr);
    return (int)pgx::ceil_div(pgx::ceil_div(n, tile), seg);
}

int pgx_blitsaw_biquad_wide_seg(float *out, int64_t out_stride, int batch, int64_t n, const double *saw_tables,
                                const double *saw_state_in, double *saw_state_out, const double *coef,
                                const double *biquad_tables, const double *biquad_state_in, double *biquad_state_out,
                                const float *gain, int64_t gain_stride, int64_t settle_frames) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && saw_tables && saw_state_in && saw_state_out && coef && biquad_tables && biquad_state_in &&
                      biquad_state_out && saw_state_in != saw_state_out && biquad_state_in != biquad_state_out,
                  "pgx_blitsaw_biquad_wide_seg: bad argument (states are read from one buffer and written to another)");
    PGX_CHECK_ARG(settle_frames > 0, "pgx_blitsaw_biquad_wide_seg: the filters' settle_frames must be known (> 0)");
    PGX_CHECK_ARG(batch == 1 || out_stride >= n, "pgx_blitsaw_biquad_wide_seg: out_stride too small");
    PGX_CHECK_ARG(gain == nullptr || batch == 1 || gain_stride >= n, "pgx_blitsaw_biquad_wide_seg: gain_stride too small");
    PGX_CHECK_ARG(batch <= 65535, "pgx_blitsaw_biquad_wide_seg: too many voices");
    constexpr int64_t tile = 4 * 64 * kSswT;
    int warm = 0;
    const int seg = bbw_segment_tiles(batch, n, settle_frames, &warm);
    const int nseg = (int)pgx::ceil_div(pgx::ceil_div(n, tile), seg);
    double *s_in = const_cast<double *>(saw_state_in), *b_in = const_cast<double *>(biquad_state_in);
    if (gain != nullptr)
        hipLaunchKernelGGL((k_blitsaw_biquad_wide<4, true, true>), dim3(batch, nseg), dim3(4 * 64), 0, pgx::stream(), out,
                           out_stride, n, saw_tables, s_in, coef, biquad_tables, b_in, gain, gain_stride, saw_state_out,
                           biquad_state_out, seg, warm);
    else
        hipLaunchKernelGGL((k_blitsaw_biquad_wide<4, false, true>), dim3(batch, nseg), dim3(4 * 64), 0, pgx::stream(), out,
                           out_stride, n, saw_tables, s_in, coef, biquad_tables, b_in, gain, gain_stride, saw_state_out,
                           biquad_state_out, seg, warm);
    PGX_LAUNCH_CHECK("k_blitsaw_biquad_wide<seg>");
    return PGX_OK;
}

// The same voices mixed on chip (k_voice_tiles): out[n] is the MixPE's block, no [voices][frames] layer in between.
struct VoiceTilesPlan {
    int emit, tiles, groups;
    int64_t stride;                       // doubles per row of partial sums
    size_t entries_bytes, bytes;
};
static VoiceTilesPlan voice_tiles_plan(int nvoices, int64_t n, int64_t warm) {
    constexpr int64_t tile = 4 * 64 * kSswT;
    VoiceTilesPlan p{};
    p.emit = (int)(tile - warm);
    p.tiles = (int)pgx::ceil_div(n, (int64_t)p.emit);
    // (voice, tile) pairs over the workgroups the chip holds at once (PGX_VT_WAVES 4-wave workgroups per CU): one round
    // of workgroups with the same number of voices (+- 1), the groups a multiple of 8 (one XCD per group: k_voice_tiles);
    // PGX_VT_GROUPS / PGX_VT_SLOTS: experiments
    static const int forced = getenv("PGX_VT_GROUPS") ? atoi(getenv("PGX_VT_GROUPS")) : 0;
    static const int slots = getenv("PGX_VT_SLOTS") ? atoi(getenv("PGX_VT_SLOTS")) : PGX_VT_WAVES * pgx::kNumCU;
    int64_t groups = forced > 0 ? forced : slots / p.tiles;
    if (groups >= 8) groups &= ~(int64_t)7;
    if (groups < 1) groups = 1;
    if (groups > nvoices) groups = nvoices;
    p.groups = (int)groups;
    p.stride = (n + 1) & ~(int64_t)1;
    p.entries_bytes = (size_t)nvoices * p.tiles * kVtEntryDoubles * sizeof(double);
    p.bytes = 2 * p.entries_bytes + (size_t)p.groups * p.stride * sizeof(double);     // two sets of entries: this block's and the next one's
    return p;
}

int64_t pgx_voice_tiles_max_warm(void) { return 2048; }

size_t pgx_voice_tiles_table_bytes(int nvoices) { return (size_t)(nvoices > 0 ? nvoices : 0) * kVtPack * sizeof(double); }

int pgx_voice_tiles_tables(double *tables, const double *saw_tables, const double *coef, const double *biquad_tables,
                           int nvoices) {
    PGX_REQUIRE_INIT();
    if (nvoices <= 0) return PGX_OK;
    PGX_CHECK_ARG(tables && saw_tables && coef && biquad_tables, "pgx_voice_tiles_tables: bad argument");
    hipLaunchKernelGGL(k_voice_tile_pack, dim3(nvoices), dim3(256), 0, pgx::stream(), tables, saw_tables, coef,
                       biquad_tables);
    PGX_LAUNCH_CHECK("k_voice_tile_pack");
    return PGX_OK;
}

size_t pgx_voice_tiles_workspace_bytes(int nvoices, int64_t n, int64_t warm_frames) {
    if (nvoices <= 0 || n <= 0 || warm_frames <= 0 || warm_frames > pgx_voice_tiles_max_warm() || (warm_frames & 15)) return 0;
    return voice_tiles_plan(nvoices, n, warm_frames).bytes;
}

int pgx_voice_tiles_entries(void *workspace, int slot, int nvoices, int64_t n, const double *tables,
                            const double *saw_state, int64_t advance_frames, int64_t warm_frames) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || nvoices <= 0) return PGX_OK;
    PGX_CHECK_ARG(workspace && tables && saw_state && (slot == 0 || slot == 1) && advance_frames >= 0,
                  "pgx_voice_tiles_entries: bad argument");
    PGX_CHECK_ARG(warm_frames > 0 && warm_frames <= pgx_voice_tiles_max_warm() && (warm_frames & 15) == 0,
                  "pgx_voice_tiles_entries: warm_frames must be a multiple of 16 in 16 .. pgx_voice_tiles_max_warm()");
    const VoiceTilesPlan p = voice_tiles_plan(nvoices, n, warm_frames);
    double *entries = reinterpret_cast<double *>(static_cast<char *>(workspace) + (size_t)slot * p.entries_bytes);
    hipLaunchKernelGGL(k_voice_tile_entries, dim3(nvoices), dim3(256), 0, pgx::stream(), entries, tables, saw_state,
                       advance_frames, p.tiles, p.emit, (int)warm_frames);
    PGX_LAUNCH_CHECK("k_voice_tile_entries");
    return PGX_OK;
}

int pgx_voice_tiles(float *out, int nvoices, int64_t n, const double *tables, const double *saw_state_in,
                    double *saw_state_out, const double *biquad_state_in, double *biquad_state_out, const float *gain,
                    int64_t gain_stride, int64_t warm_frames, void *workspace, int entries_slot, int next_entries_slot) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || nvoices <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && tables && saw_state_in && saw_state_out && biquad_state_in && biquad_state_out && workspace &&
                      saw_state_in != saw_state_out && biquad_state_in != biquad_state_out,
                  "pgx_voice_tiles: bad argument (states are read from one buffer and written to another)");
    PGX_CHECK_ARG(warm_frames > 0 && warm_frames <= pgx_voice_tiles_max_warm() && (warm_frames & 15) == 0,
                  "pgx_voice_tiles: warm_frames must be a multiple of 16 in 16 .. pgx_voice_tiles_max_warm()");
    PGX_CHECK_ARG(gain == nullptr || nvoices == 1 || gain_stride >= n, "pgx_voice_tiles: gain_stride too small");
    PGX_CHECK_ARG(n < ((int64_t)1 << 40), "pgx_voice_tiles: block too long");
    const VoiceTilesPlan p = voice_tiles_plan(nvoices, n, warm_frames);
    PGX_CHECK_ARG((int64_t)p.tiles * p.groups < ((int64_t)1 << 31), "pgx_voice_tiles: too many tiles");
    PGX_CHECK_ARG(entries_slot >= -1 && entries_slot <= 1, "pgx_voice_tiles: entries_slot is -1 (made here), 0 or 1");
    PGX_CHECK_ARG(next_entries_slot >= -1 && next_entries_slot <= 1 && (next_entries_slot < 0 || next_entries_slot != (entries_slot < 0 ? 0 : entries_slot)),
                  "pgx_voice_tiles: next_entries_slot is -1 (none) or the set this block does not read");
    if (entries_slot < 0) {
        const int rc = pgx_voice_tiles_entries(workspace, 0, nvoices, n, tables, saw_state_in, 0, warm_frames);
        if (rc != PGX_OK) return rc;
        entries_slot = 0;
    }
    double *entries = reinterpret_cast<double *>(static_cast<char *>(workspace) + (size_t)entries_slot * p.entries_bytes);
    double *partial = reinterpret_cast<double *>(static_cast<char *>(workspace) + 2 * p.entries_bytes);
    const dim3 grid((unsigned)(p.tiles * p.groups));
    if (gain != nullptr)
        hipLaunchKernelGGL((k_voice_tiles<4, true>), grid, dim3(256), 0, pgx::stream(), partial, p.stride, n, nvoices,
                           p.groups, p.tiles, p.emit, (int)warm_frames, tables, saw_state_in, saw_state_out,
                           biquad_state_in, biquad_state_out, (const double *)entries, gain, gain_stride);
    else
        hipLaunchKernelGGL((k_voice_tiles<4, false>), grid, dim3(256), 0, pgx::stream(), partial, p.stride, n, nvoices,
                           p.groups, p.tiles, p.emit, (int)warm_frames, tables, saw_state_in, saw_state_out,
                           biquad_state_in, biquad_state_out, (const double *)entries, gain, gain_stride);
    PGX_LAUNCH_CHECK("k_voice_tiles");
    const unsigned mix_blocks = (unsigned)pgx::ceil_div(n, (int64_t)256);
    if (next_entries_slot >= 0) {
        double *next = reinterpret_cast<double *>(static_cast<char *>(workspace) + (size_t)next_entries_slot * p.entries_bytes);
        hipLaunchKernelGGL(k_mix_partials_entries, dim3(mix_blocks + (unsigned)nvoices), dim3(256), 0, pgx::stream(), out,
                           (const double *)partial, p.stride, p.groups, n, nvoices, next, tables,
                           (const double *)saw_state_out, p.tiles, p.emit, (int)warm_frames);
    } else {
        hipLaunchKernelGGL(k_mix_partials, dim3(mix_blocks), dim3(256), 0, pgx::stream(), out, (const double *)partial,
                           p.stride, p.groups, n);
    }
    PGX_LAUNCH_CHECK("k_mix_partials");
    return PGX_OK;
}

int pgx_supersaw_bank(float *out, int64_t out_stride, int batch, int nvoices, int64_t n, int channels,
                      double sample_rate, const pgx_blitsaw_params *params, double *state,
                      const double *amp_scalar) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && params && state && amp_scalar && channels >= 1 && sample_rate > 0,
                  "pgx_supersaw_bank: bad argument");
    PGX_CHECK_ARG(nvoices >= 1 && nvoices <= 16, "pgx_supersaw_bank: 1..16 voices per instance");
    PGX_CHECK_ARG(batch == 1 || out_stride >= n * channels, "pgx_supersaw_bank: out_stride too small");
    // (a third wave per SIMD, forced with a 168-VGPR cap, spills and is slower: 0.88 against 0.78 ms per block)
    const int whole = 1 << 30;             // one segment: the whole block
    if (batch > 256)                       // two 4-wave workgroups per CU; up to 256 instances: one 8-wave each
        hipLaunchKernelGGL(k_supersaw_bank<4>, dim3(batch), dim3(4 * 64), 0, pgx::stream(), out, out_stride,
                           nvoices, n, channels, sample_rate, params, (const double *)state, state, amp_scalar, whole,
                           (const double *)nullptr);
    else
        hipLaunchKernelGGL(k_supersaw_bank<8>, dim3(batch), dim3(8 * 64), 0, pgx::stream(), out, out_stride,
                           nvoices, n, channels, sample_rate, params, (const double *)state, state, amp_scalar, whole,
                           (const double *)nullptr);
    PGX_LAUNCH_CHECK("k_supersaw_bank");
    return PGX_OK;
}

// The segmented bank: 8-wave workgroups on 4096-frame tiles, one per CU, or 4-wave workgroups on 2048-frame tiles,
// two per CU (one's memory latencies and barrier waits overlap the other's arithmetic).  PGX_SS_SEG_NW picks (experiments).
static int ss_seg_nw() {
    static const int nw = getenv("PGX_SS_SEG_NW") ? atoi(getenv("PGX_SS_SEG_NW")) : 8;
    return nw == 4 ? 4 : 8;
}

int pgx_supersaw_bank_segments(int batch, int64_t n) {
    if (batch <= 0 || n <= 0 || batch > 256) return 1;
    const int nw = ss_seg_nw();
    const int64_t tiles = pgx::ceil_div(n, nw * 64 * kSawT);
    int64_t want = pgx::ceil_div((nw == 4 ? 2 : 1) * pgx::kNumCU, batch);
    static const int forced = getenv("PGX_SS_SEGS") ? atoi(getenv("PGX_SS_SEGS")) : 0;      // experiments
    if (forced > 0) want = forced;
    if (want > tiles) want = tiles;
    if (want < 1) want = 1;
    const int64_t seg_tiles = pgx::ceil_div(tiles, want);
    return (int)pgx::ceil_div(tiles, seg_tiles);
}

size_t pgx_supersaw_bank_table_bytes(int batch, int nvoices) {
    if (batch <= 0 || nvoices <= 0) return 0;
    return (size_t)batch * nvoices * kSsTabDoubles * sizeof(double);
}

int pgx_supersaw_bank_tables(double *tables, int batch, int nvoices, double sample_rate,
                             const pgx_blitsaw_params *params) {
    PGX_REQUIRE_INIT();
    if (batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(tables && params && nvoices >= 1 && nvoices <= 16 && sample_rate > 0,
                  "pgx_supersaw_bank_tables: bad argument");
    hipLaunchKernelGGL(k_supersaw_tables, dim3(batch), dim3(256), 0, pgx::stream(), tables, nvoices, sample_rate,
                       params);
    PGX_LAUNCH_CHECK("k_supersaw_tables");
    return PGX_OK;
}

int pgx_supersaw_bank_seg(float *out, int64_t out_stride, int batch, int nvoices, int64_t n, int channels,
                          double sample_rate, const pgx_blitsaw_params *params, const double *state_in,
                          double *state_out, const double *amp_scalar, const double *tables) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && params && state_in && state_out && state_in != state_out && amp_scalar && channels >= 1 &&
                      sample_rate > 0,
                  "pgx_supersaw_bank_seg: bad argument (state_in and state_out must be two buffers)");
    PGX_CHECK_ARG(nvoices >= 1 && nvoices <= 16, "pgx_supersaw_bank_seg: 1..16 voices per instance");
    PGX_CHECK_ARG(batch == 1 || out_stride >= n * channels, "pgx_supersaw_bank_seg: out_stride too small");
    if (batch > 256) {                     // enough instances for two 4-wave workgroups per CU: one segment each
        hipLaunchKernelGGL(k_supersaw_bank<4>, dim3(batch), dim3(4 * 64), 0, pgx::stream(), out, out_stride, nvoices,
                           n, channels, sample_rate, params, state_in, state_out, amp_scalar, 1 << 30, tables);
        PGX_LAUNCH_CHECK("k_supersaw_bank");
        return PGX_OK;
    }
    const int nw = ss_seg_nw();
    const int64_t tiles = pgx::ceil_div(n, nw * 64 * kSawT);
    const int nseg = pgx_supersaw_bank_segments(batch, n);
    const int seg_tiles = (int)pgx::ceil_div(tiles, nseg);
    if (nw == 4)
        hipLaunchKernelGGL(k_supersaw_bank<4>, dim3(batch, nseg), dim3(4 * 64), 0, pgx::stream(), out, out_stride,
                           nvoices, n, channels, sample_rate, params, state_in, state_out, amp_scalar, seg_tiles, tables);
    else
        hipLaunchKernelGGL(k_supersaw_bank<8>, dim3(batch, nseg), dim3(8 * 64), 0, pgx::stream(), out, out_stride,
                           nvoices, n, channels, sample_rate, params, state_in, state_out, amp_scalar, seg_tiles, tables);
    PGX_LAUNCH_CHECK("k_supersaw_bank<segments>");
    return PGX_OK;
}

// k_supersaw_wide: 4-wave workgroups on 4096-frame tiles (16 frames per thread).
size_t pgx_supersaw_wide_table_bytes(int batch, int nvoices) {
    if (batch <= 0 || nvoices <= 0) return 0;
    return (size_t)batch * nvoices * kSswTabDoubles * sizeof(double);
}

int pgx_supersaw_wide_tables(double *tables, int batch, int nvoices, double sample_rate,
                             const pgx_blitsaw_params *params) {
    PGX_REQUIRE_INIT();
    if (batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(tables && params && nvoices >= 1 && nvoices <= kSsMaxVoices && sample_rate > 0,
                  "pgx_supersaw_wide_tables: bad argument");
    hipLaunchKernelGGL(k_supersaw_wide_tables, dim3(batch), dim3(64), 0, pgx::stream(), tables, nvoices, sample_rate,
                       params);
    PGX_LAUNCH_CHECK("k_supersaw_wide_tables");
    return PGX_OK;
}

// How many time segments: the count with the smallest estimated makespan.  Measured on MI355X (tools/ssw_probe.py,
// tools/microbench/ss_phases.hip): a 4096-frame tile of 7 voices takes a workgroup 9.8 us when it has its CU to
// itself (one wave per SIMD), 15.4 us for two workgroups sharing a CU (x1.57), ~x2.25 for three, x0.75 per
// workgroup beyond (with the anchors kept in LDS a tile after the first is 7.9 us: 1.13 us per voice); entering a
// segment costs the tables (0.9 us), the closed-form carries (1.9 us per round of four voices) and the anchor sines of
// its first tile (0.28 us per voice) -- 0.85 of a tile for 7 voices, 2.7 tiles for a lone oscillator; taken a
// quarter higher (a plan with more segments has to win clearly).
int pgx_supersaw_wide_segments(int batch, int nvoices, int64_t n) {
    if (batch <= 0 || n <= 0 || nvoices <= 0) return 1;
    const double entry = 1.25 * (0.9 + 1.9 * (double)((nvoices + 3) / 4) + 0.28 * (double)nvoices) / (1.13 * (double)nvoices);
    constexpr int64_t tile = 4 * 64 * kSswT;
    const int64_t tiles = pgx::ceil_div(n, tile);
    static const int forced = getenv("PGX_SSW_SEGS") ? atoi(getenv("PGX_SSW_SEGS")) : 0;      // experiments
    int64_t best = 1;
    double best_cost = 0.0;
    for (int64_t k = 1; k <= tiles && k <= 4096; ++k) {        // (grid.y)
        const int64_t seg_tiles = pgx::ceil_div(tiles, k);
        const int64_t nseg = pgx::ceil_div(tiles, seg_tiles);
        if (nseg != k) continue;                                // (the same plan as a smaller k)
        const int64_t per_cu = pgx::ceil_div((int64_t)batch * nseg, pgx::kNumCU);
        const double share = per_cu <= 1 ? 1.0 : per_cu == 2 ? 1.57 : per_cu == 3 ? 2.25 : 0.75 * (double)per_cu;
        const double cost = ((double)seg_tiles + (nseg > 1 ? entry : 0.1)) * share;
        if (k == 1 || cost < 0.97 * best_cost) {                // (more segments have to win clearly)
            best = k;
            best_cost = cost;
        }
        if (forced > 0 && k == forced) return (int)k;
    }
    return (int)best;
}

int pgx_supersaw_wide(float *out, int64_t out_stride, int batch, int nvoices, int64_t n, int channels,
                      const double *state_in, double *state_out, const double *amp_scalar, const double *tables) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && state_in && state_out && state_in != state_out && amp_scalar && tables && channels >= 1,
                  "pgx_supersaw_wide: bad argument (state_in and state_out must be two buffers)");
    PGX_CHECK_ARG(nvoices >= 1 && nvoices <= kSsMaxVoices, "pgx_supersaw_wide: 1..16 voices per instance");
    PGX_CHECK_ARG(batch == 1 || out_stride >= n * channels, "pgx_supersaw_wide: out_stride too small");
    PGX_CHECK_ARG(batch <= 65535, "pgx_supersaw_wide: too many instances");
    constexpr int64_t tile = 4 * 64 * kSswT;
    const int64_t tiles = pgx::ceil_div(n, tile);
    const int nseg = pgx_supersaw_wide_segments(batch, nvoices, n);
    const int seg_tiles = (int)pgx::ceil_div(tiles, nseg);
    const int keep_anchors = nvoices <= kSswAnchorVoices ? 1 : 0;
    // dynamic LDS: the voices' tables (1.7 KB each) and, for up to 8 voices, the threads' anchors (8 KB per voice): 69 KB
    // for 7 voices -- two workgroups per CU
    const size_t lds = (size_t)nvoices * kSswTabDoubles * sizeof(double) +
                       (keep_anchors ? (size_t)nvoices * 4 * 64 * sizeof(SswAnchor) : 0);
    // (once: the call costs tens of microseconds of host time -- per launch it made a rank's 64-instance block
    // host-bound, 48 -> 86 us)
    static bool lds_allowed = false;
    if (!lds_allowed) {
        const size_t most = (size_t)kSswAnchorVoices * (kSswTabDoubles * sizeof(double) + 4 * 64 * sizeof(SswAnchor));
        const size_t plain = (size_t)kSsMaxVoices * kSswTabDoubles * sizeof(double);
        PGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_supersaw_wide<4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(most > plain ? most : plain)));
        lds_allowed = true;
    }
    hipLaunchKernelGGL(k_supersaw_wide<4>, dim3(batch, nseg), dim3(4 * 64), lds, pgx::stream(), out, out_stride,
                       nvoices, n, channels, state_in, state_out, amp_scalar, seg_tiles, tables, keep_anchors);
    PGX_LAUNCH_CHECK("k_supersaw_wide");
    return PGX_OK;
}

int pgx_gate_stateful(float *out, int64_t n, double sample_rate, double freq, double duty, double phase,
                      const float *freq_stream, const float *duty_stream, const float *phase_stream, double *state) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && state && sample_rate > 0, "pgx_gate_stateful: bad argument");
    hipLaunchKernelGGL(k_gate_stateful, dim3(1), dim3(kBlock), 0, pgx::stream(), out, n, sample_rate, freq, duty,
                       phase, freq_stream, duty_stream, phase_stream, state);
    PGX_LAUNCH_CHECK("k_gate_stateful");
    return PGX_OK;
}

int pgx_sine_stateful(float *out, int64_t n, int channels, double sample_rate,
                      const pgx_sine_stateful_params *params, const float *freq, const float *amp,
                      const float *phase_mod, double *state) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && params && state && channels >= 1 && sample_rate > 0, "pgx_sine_stateful: bad argument");
    hipLaunchKernelGGL(k_sine_stateful, dim3(1), dim3(kBlock), 0, pgx::stream(), out, n, channels, sample_rate,
                       params, freq, amp, phase_mod, state);
    PGX_LAUNCH_CHECK("k_sine_stateful");
    return PGX_OK;
}


int pgx_svf(float *out, const float *in, int64_t n, int channels, double sample_rate,
            const pgx_biquad_var_params *params, const float *freq, const float *q, double gain_a,
            const double *coef, double *state, void *workspace) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && params && state && channels >= 1 && sample_rate > 0, "pgx_svf: bad argument");
    PGX_CHECK_ARG(!(coef && (freq || q)), "pgx_svf: constant coefficients exclude control streams");
    const Scan2Plan p = scan2_plan(n, kSvTile);
    double *snap = (double *)workspace;
    double *agg = snap ? snap + (size_t)channels * 4 : nullptr;
    if (p.nseg <= 1 || workspace == nullptr) {
        hipLaunchKernelGGL(k_svf<2>, dim3(1, channels), dim3(kBlock), 0, pgx::stream(), out, in, n, channels,
                           sample_rate, params, freq, q, gain_a, coef, state, snap, agg,
                           (int)pgx::ceil_div(n, kSvTile), 1);
        PGX_LAUNCH_CHECK("k_svf");
        return PGX_OK;
    }
    hipLaunchKernelGGL(k_svf<0>, dim3(p.nseg, channels), dim3(kBlock), 0, pgx::stream(), out, in, n, channels,
                       sample_rate, params, freq, q, gain_a, coef, state, snap, agg, p.seg_tiles, p.nseg);
    PGX_LAUNCH_CHECK("k_svf<reduce>");
    hipLaunchKernelGGL(k_svf<1>, dim3(p.nseg, channels), dim3(kBlock), 0, pgx::stream(), out, in, n, channels,
                       sample_rate, params, freq, q, gain_a, coef, state, snap, agg, p.seg_tiles, p.nseg);
    PGX_LAUNCH_CHECK("k_svf<apply>");
    return PGX_OK;
}

constexpr int64_t kEnvMwMinFrames = 16 * (int64_t)kEnvMwWindow;      // from here on all windows at once

size_t pgx_envelope_scratch_bytes(int64_t n, int channels) {
    if (n <= 0 || channels <= 0) return 0;
    const size_t base = (size_t)(n + (n + kEnvBlock - 1) / kEnvBlock) * channels;
    const size_t nwin = (size_t)((n + kEnvMwWindow - 1) / kEnvMwWindow);
    return (base + 5 * nwin * channels + 8) * sizeof(double);              // + pieces (2 x 2), entries, control
}

int pgx_envelope(float *out, const float *in, int64_t n, int channels, double attack_coeff,
                 double release_coeff, int one_pole, int rms_window, int64_t rms_period, double *state,
                 double *scratch) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && state && scratch && channels >= 1, "pgx_envelope: bad argument");
    const bool fused_peak = !one_pole && rms_window <= 0;                   // |x| is taken inside the follower
    if (!fused_peak) {
        double *blocks = scratch + n * channels;                            // after the detector output
        if (rms_window > 0) {
            const int64_t items = (n + kEnvBlock - 1) / kEnvBlock * channels;
            hipLaunchKernelGGL(k_env_blocks, dim3((unsigned)((items + kWaves - 1) / kWaves)), dim3(kBlock), 0,
                               pgx::stream(), blocks, in, n, channels);
            PGX_LAUNCH_CHECK("k_env_blocks");
        }
        hipLaunchKernelGGL(k_env_detect, dim3(pgx::grid_for(n * channels, kBlock)), dim3(kBlock), 0, pgx::stream(),
                           scratch, in, blocks, n, channels, rms_window, rms_period);
        PGX_LAUNCH_CHECK("k_env_detect");
    }
    if (one_pole) {
        hipLaunchKernelGGL(k_env_onepole, dim3(channels), dim3(kBlock), 0, pgx::stream(), out,
                           (const double *)scratch, n, channels, attack_coeff, state);
        PGX_LAUNCH_CHECK("k_env_onepole");
    } else {
        const double *det = fused_peak ? nullptr : (const double *)scratch;
        const int *run_flag = nullptr;
        const int64_t nwin64 = (n + kEnvMwWindow - 1) / kEnvMwWindow;
        if (n >= kEnvMwMinFrames && nwin64 <= 65535 && channels <= 65535) {
            // all windows at once, one launch per Newton round over the windows' entry levels
            const int nwin = (int)nwin64;
            double *mw = scratch + (size_t)(n + (n + kEnvBlock - 1) / kEnvBlock) * channels;
            double *pa_buf = mw, *pb_buf = mw + 2 * (size_t)nwin * channels, *guess = mw + 4 * (size_t)nwin * channels;
            EnvMwCtl *ctl = reinterpret_cast<EnvMwCtl *>(mw + 5 * (size_t)nwin * channels);
            if (int rc = pgx_memset(ctl, 0, sizeof(EnvMwCtl))) return rc;
            // (PGX_ENV_MW_ROUNDS=1 makes every block give up: the test of the fallback)
            static const int rounds_env = getenv("PGX_ENV_MW_ROUNDS") ? atoi(getenv("PGX_ENV_MW_ROUNDS")) : kEnvMwRounds;
            const int rounds = rounds_env < 1 ? 1 : (rounds_env > kEnvMwRounds ? kEnvMwRounds : rounds_env);
            for (int r = 0; r < rounds; ++r) {
                if (fused_peak)
                    hipLaunchKernelGGL(k_env_newton_mw<true>, dim3(nwin, channels), dim3(kEnvMwNW * 64), 0,
                                       pgx::stream(), out, in, det, n, channels, attack_coeff, release_coeff,
                                       (const double *)state, r, nwin, pa_buf, pb_buf, guess, ctl);
                else
                    hipLaunchKernelGGL(k_env_newton_mw<false>, dim3(nwin, channels), dim3(kEnvMwNW * 64), 0,
                                       pgx::stream(), out, in, det, n, channels, attack_coeff, release_coeff,
                                       (const double *)state, r, nwin, pa_buf, pb_buf, guess, ctl);
                PGX_LAUNCH_CHECK("k_env_newton_mw");
            }
            hipLaunchKernelGGL(k_env_mw_finish, dim3(channels), dim3(64), 0, pgx::stream(), state, channels, nwin,
                               (const double *)pa_buf, (const double *)pb_buf, ctl, rounds);
            PGX_LAUNCH_CHECK("k_env_mw_finish");
            run_flag = &ctl->fallback;                // the sequential kernel below runs only if that gave up
        }
#define PGX_ENV_LAUNCH(NW, T, PEAK)                                                                               \
        hipLaunchKernelGGL((k_env_newton<NW, T, PEAK>), dim3(channels), dim3(NW * 64), 0, pgx::stream(), out, in, det, \
                           n, channels, attack_coeff, release_coeff, state, run_flag)
        if (n <= 1024) {
            if (fused_peak) PGX_ENV_LAUNCH(4, 4, true);
            else PGX_ENV_LAUNCH(4, 4, false);
        } else if (n <= 2048) {
            if (fused_peak) PGX_ENV_LAUNCH(8, 4, true);
            else PGX_ENV_LAUNCH(8, 4, false);
        } else if (n <= 4096) {
            if (fused_peak) PGX_ENV_LAUNCH(8, 8, true);
            else PGX_ENV_LAUNCH(8, 8, false);
        } else {                                     // 8192-sample windows: fewest rounds x windows (measured)
            if (fused_peak) PGX_ENV_LAUNCH(8, 16, true);
            else PGX_ENV_LAUNCH(8, 16, false);
        }
#undef PGX_ENV_LAUNCH
        PGX_LAUNCH_CHECK("k_env_newton");
    }
    return PGX_OK;
}

int pgx_transform(float *out, const float *in, int64_t n_elems, const pgx_transform_op *ops, int nops) {
    PGX_REQUIRE_INIT();
    if (n_elems <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && (ops || nops == 0) && nops >= 0, "pgx_transform: bad argument");
    hipLaunchKernelGGL(k_transform, dim3(pgx::grid_for(n_elems, kBlock)), dim3(kBlock), 0, pgx::stream(), out,
                       in, n_elems, ops, nops);
    PGX_LAUNCH_CHECK("k_transform");
    return PGX_OK;
}

}  // extern "C"
