// pgx_seq.hip -- PEs whose recurrences are not associative scans: LadderPE (nonlinear) and
// CombPE (integer-indexed delay line).  (The ADSR state machines live in pgx_adsr.hip.)
//
// Parallelism comes from independent chains (voices x channels); inside one chain the
// reference's per-sample operation order is followed literally (-ffp-contract=off).

#include <cstdlib>

#include "pgx_common.h"

namespace {

// ================================================================================================
// LadderPE (ladder_pe.py:31-203).
//
// The recurrence is nonlinear (tanh inside the feedback loop), so a chain advances one sample at a
// time in the reference's exact float64 operation order: ladder_advance().
//   * k_ladder: one lane per (instance, channel) chain over the whole block -- exact, but the whole
//     GPU then runs `chains` lanes.
//   * k_ladder_segments + k_ladder_finish: time-parallel.  Below self-oscillation the ladder is a
//     contraction: two trajectories driven by the same input converge geometrically whatever their
//     starting states.  The block is cut into segments; every segment but those reaching the block
//     start begins `settle` samples early from a zero state and discards that warm-up output.  The
//     state each segment reached at its first output sample is then compared with the state its left
//     neighbour ended in (k_ladder_finish, one wave per chain); on any disagreement beyond 1e-8
//     relative the chain is simply re-rendered sequentially from the carried state, so the result
//     never depends on the convergence assumption, only the speed does.
// ================================================================================================
struct LadderConsts {
    double sr, pbg, oversample_recip, max_cutoff;
    int oversample, mode, channels, ch;
    const float *x, *freq, *resonance, *drive;
    float *o;
    double p_freq, p_res, p_drive;
};

struct LadderState {
    double z0[4], z1[4], old_input;
};

__device__ __forceinline__ double ladder_drive_scale(double drv, double pbg) {
    if (drv < 0.0) drv = 0.0;
    if (drv > 1.0) {
        if (drv > 4.0) drv = 4.0;
        return 1.0 + (drv - 1.0) * (1.0 - pbg);
    }
    return drv;
}

__device__ __forceinline__ LadderConsts ladder_consts(const pgx_ladder_params &p, double sr, int channels, int ch,
                                                      const float *x, float *o, const float *freq,
                                                      const float *resonance, const float *drive) {
    LadderConsts c;
    c.sr = sr;
    c.pbg = p.passband_gain;
    c.oversample = p.oversample;
    c.oversample_recip = 1.0 / (double)p.oversample;
    const double nyquist = sr / 2.0;
    c.max_cutoff = nyquist * 0.85;
    if (c.max_cutoff > nyquist - 1.0) c.max_cutoff = nyquist - 1.0;
    c.mode = p.mode;
    c.channels = channels;
    c.ch = ch;
    c.x = x; c.o = o; c.freq = freq; c.resonance = resonance; c.drive = drive;
    c.p_freq = p.freq; c.p_res = p.resonance; c.p_drive = p.drive;
    return c;
}

__device__ __forceinline__ void ladder_store(double *st, const LadderState &s);

constexpr int kLadderChunk = 8;

// A lane walks its own stretch of the stream, so a wave's loads are 64 different cache lines whatever their width:
// what counts is the number of load instructions (each occupies the CU's one address unit for 64 cycles).  Mono
// streams are fetched and stored 16 bytes at a time (two instructions per chunk instead of eight); the addresses
// are only 4-byte aligned, which global memory instructions accept.
struct __attribute__((packed, aligned(4))) LadderF4 {
    float v[4];
};
__device__ __forceinline__ void ladder_load8(const LadderConsts &c, int64_t base, float (&x)[kLadderChunk]) {
    if (c.channels == 1) {
        const LadderF4 a = *reinterpret_cast<const LadderF4 *>(c.x + base);
        const LadderF4 b = *reinterpret_cast<const LadderF4 *>(c.x + base + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[j] = a.v[j];
            x[4 + j] = b.v[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < kLadderChunk; ++j) x[j] = c.x[(base + j) * c.channels + c.ch];
    }
}
__device__ __forceinline__ void ladder_store8(const LadderConsts &c, int64_t base, const float (&y)[kLadderChunk]) {
    if (c.channels == 1) {
        LadderF4 a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a.v[j] = y[j];
            b.v[j] = y[4 + j];
        }
        *reinterpret_cast<LadderF4 *>(c.o + base) = a;
        *reinterpret_cast<LadderF4 *>(c.o + base + 4) = b;
    } else {
#pragma unroll
        for (int j = 0; j < kLadderChunk; ++j) c.o[(base + j) * c.channels + c.ch] = y[j];
    }
}

// Samples [i0, i1); output is written from `emit_from` on; when `snap` is given the state on entering
// sample `emit_from` is stored there (i0 <= emit_from <= i1).
// Input is fetched a chunk ahead and output stored a chunk at a time: the recurrence is latency
// bound, one memory round trip per sample would double its run time.
// STREAMS = false: scalar cutoff / resonance / drive -- the coefficient polynomials leave the sample loop (the
// chain is one wave per SIMD, so every instruction counts, not only the dependent ones).  OS = 2: the usual
// oversampling factor, unrolled; OS = 0: any factor.  UMODE: every lane of the wave has the same LadderMode, kept in
// an SGPR so that the output tap is chosen by scalar branches.  Same operations in the same order either way.
template <bool STREAMS, int OS, bool UMODE>
__device__ __forceinline__ void ladder_advance_impl(const LadderConsts &c, LadderState &s, int64_t i0,
                                                    int64_t emit_from, int64_t i1, double *snap, int mode_u) {
    const int mode = UMODE ? mode_u : c.mode;
    const double state_decay = 0.95, input_threshold = 1e-5, resonance_multiplier = 1.8;
    const double two_pi = 2.0 * 3.141592653589793;
    const double min_cutoff = 5.0;
    const int oversample = OS ? OS : c.oversample;
    double alpha = 0.0, q_adjust = 0.0, k = 0.0, drive_scaled = 0.0;
    auto coefficients = [&](int64_t i) {
        double cutoff = (STREAMS && c.freq) ? (double)c.freq[i] : c.p_freq;
        if (cutoff < min_cutoff) cutoff = min_cutoff;
        if (cutoff > c.max_cutoff) cutoff = c.max_cutoff;
        const double wc = cutoff * two_pi / (c.sr * (double)c.oversample);
        const double wc2 = wc * wc;
        const double wc3 = wc2 * wc;
        const double wc4 = wc3 * wc;
        alpha = 0.9892 * wc - 0.4324 * wc2 + 0.1381 * wc3 - 0.0202 * wc4;
        q_adjust = 1.006 + 0.0536 * wc - 0.095 * wc2 - 0.05 * wc4;
        double res = (STREAMS && c.resonance) ? (double)c.resonance[i] : c.p_res;
        if (res < 0.0) res = 0.0;
        if (res > 1.0) res = 1.0;
        k = 4.0 * res * resonance_multiplier;
        drive_scaled = ladder_drive_scale((STREAMS && c.drive) ? (double)c.drive[i] : c.p_drive, c.pbg);
    };
    if (!STREAMS) coefficients(0);

    // one sample: the reference's operation order (ladder_pe.py:116-203)
    auto sample = [&](int64_t i, float x) -> float {
        if (STREAMS) coefficients(i);
        const double input_sample = (double)x * drive_scaled;
        const double input_abs = input_sample >= 0.0 ? input_sample : -input_sample;
        if (input_abs < input_threshold) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s.z0[q] *= state_decay;
                s.z1[q] *= state_decay;
            }
            s.old_input *= state_decay;
        }
        double total = 0.0, interp = 0.0;
        for (int os = 0; os < oversample; ++os) {                    // (a constant trip count when OS != 0)
            const double in_interp = interp * s.old_input + (1.0 - interp) * input_sample;
            const double u = pgx::pgx_tanh(in_interp - (s.z1[3] - c.pbg * in_interp) * k * q_adjust);
            double ft, stage1, stage2, stage3, stage4, weighted;

            ft = u * 0.76923077 + 0.23076923 * s.z0[0] - s.z1[0];
            ft = ft * alpha + s.z1[0];
            s.z1[0] = ft; s.z0[0] = u; stage1 = ft;

            ft = stage1 * 0.76923077 + 0.23076923 * s.z0[1] - s.z1[1];
            ft = ft * alpha + s.z1[1];
            s.z1[1] = ft; s.z0[1] = stage1; stage2 = ft;

            ft = stage2 * 0.76923077 + 0.23076923 * s.z0[2] - s.z1[2];
            ft = ft * alpha + s.z1[2];
            s.z1[2] = ft; s.z0[2] = stage2; stage3 = ft;

            ft = stage3 * 0.76923077 + 0.23076923 * s.z0[3] - s.z1[3];
            ft = ft * alpha + s.z1[3];
            s.z1[3] = ft; s.z0[3] = stage3; stage4 = ft;

            if (mode == 0) weighted = stage4;
            else if (mode == 1) weighted = stage2;
            else if (mode == 2) weighted = (stage2 + stage4) * 4.0 - stage3 * 8.0;
            else if (mode == 3) weighted = (stage1 - stage2) * 2.0;
            else if (mode == 4) weighted = u + stage4 - (stage1 + stage3) * 4.0 + stage2 * 6.0;
            else weighted = u + stage2 - stage1 * 2.0;

            total += weighted * c.oversample_recip;
            interp += c.oversample_recip;
        }
        s.old_input = input_sample;
        return (float)total;
    };

    // [a, b) in whole chunks (input fetched a chunk ahead, output stored a chunk at a time, no per-sample
    // bounds tests) and a tail of single samples
    auto run = [&](int64_t a, int64_t b, bool emit) {
        int64_t base = a;
        if (b - a >= kLadderChunk) {
            float xn[kLadderChunk];
            ladder_load8(c, base, xn);
            for (; base + kLadderChunk <= b; base += kLadderChunk) {
                float xc[kLadderChunk], yc[kLadderChunk];
#pragma unroll
                for (int j = 0; j < kLadderChunk; ++j) xc[j] = xn[j];
                if (base + 2 * kLadderChunk <= b) ladder_load8(c, base + kLadderChunk, xn);
#pragma unroll
                for (int j = 0; j < kLadderChunk; ++j) yc[j] = sample(base + j, xc[j]);
                if (emit) ladder_store8(c, base, yc);
            }
        }
        for (; base < b; ++base) {
            const float y = sample(base, c.x[base * c.channels + c.ch]);
            if (emit) c.o[base * c.channels + c.ch] = y;
        }
    };
    run(i0, emit_from, false);                                   // warm-up (nothing written)
    if (snap) ladder_store(snap, s);                             // state on entering the first output sample
    run(emit_from, i1, true);
}

// Warm-up of a time segment (scalar parameters): samples [i0, i1) advance the state and nothing is written.  What
// the warm-up has to deliver is a state within ~1e-10 of the true trajectory at its end, not the reference's
// roundings along the way, so it runs on fused multiply-adds with the stages regrouped and skips the output taps.
// FAST takes tanh from the float32 exponential unit (absolute error ~1e-7, 8 dependent operations instead of 35):
// the bulk of a warm-up only has to forget the zero start; the accurate tail (tanh to ~1e-11, ladder_tanh_mid)
// then contracts the 1e-7 the fast part leaves behind.
__device__ __forceinline__ double ladder_tanh_fast(double g) {
    const float e = __expf(2.0f * (float)g);                       // inf / 0 at the ends: the quotient saturates
    return (double)__builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);   // v_rcp_f32: 1 ulp is plenty
}

// tanh to ~1e-11 absolute for the accurate tail of a warm-up (its target is 2e-10): pgx_tanh's scheme with a
// degree-9 polynomial and one Newton step on the reciprocal -- 20 dependent operations instead of 35.
__device__ __forceinline__ double ladder_tanh_mid(double x) {
    double ax = fabs(x);
    ax = ax < 40.0 ? ax : 40.0;
    const double t = ax + ax;
    const double kf = rint(t * 1.4426950408889634);
    double r = __builtin_fma(-kf, 6.93147180369123816490e-01, t);
    r = __builtin_fma(-kf, 1.90821492927058770002e-10, r);       // r in [-ln2/2, ln2/2]
    double p = -1.0 / 362880.0;                                  // exp(-r), Taylor to r^9: 7e-12 relative
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, -1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, -1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, -1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, -1.0);
    p = __builtin_fma(p, r, 1.0);
    const double e = ldexp(p, -(int)kf);                         // exp(-2|x|)
    const double den = 1.0 + e;
    double y = __builtin_amdgcn_rcp(den);
    y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
    return copysign((1.0 - e) * y, x);
}

// STREAMS: cutoff / resonance / drive per sample from the control streams (the coefficient polynomials then sit in
// the sample loop: ~15 more operations, none of them on the feedback chain).
// EMIT: the same arithmetic WITH the output taps, samples written (a time segment's own samples in k_ladder_segments:
// what a segment emits is within ~1e-10 of the reference's trajectory anyway -- it started from a warm-up, not from
// the true state -- so the fused, regrouped stages and the ~1e-11 tanh cost nothing that the form had not given up
// already, and a sample is 38 dependent operations instead of 60).
template <bool FAST, int OS, bool STREAMS = false, bool EMIT = false>
__device__ __forceinline__ void ladder_warm_impl(const LadderConsts &c, LadderState &s, int64_t i0, int64_t i1) {
    const double state_decay = 0.95, input_threshold = 1e-5, resonance_multiplier = 1.8;
    const double two_pi = 2.0 * 3.141592653589793;
    const double wc_scale = two_pi / (c.sr * (double)c.oversample);
    double kq = 0.0, drive_scaled = 0.0, ac0 = 0.0, ac1 = 0.0, oma = 0.0;
    auto coefficients = [&](int64_t i) {
        double cutoff = (STREAMS && c.freq) ? (double)c.freq[i] : c.p_freq;
        if (cutoff < 5.0) cutoff = 5.0;
        if (cutoff > c.max_cutoff) cutoff = c.max_cutoff;
        const double wc = STREAMS ? cutoff * wc_scale : cutoff * two_pi / (c.sr * (double)c.oversample);
        const double wc2 = wc * wc, wc3 = wc2 * wc, wc4 = wc3 * wc;
        const double alpha = 0.9892 * wc - 0.4324 * wc2 + 0.1381 * wc3 - 0.0202 * wc4;
        const double q_adjust = 1.006 + 0.0536 * wc - 0.095 * wc2 - 0.05 * wc4;
        double res = (STREAMS && c.resonance) ? (double)c.resonance[i] : c.p_res;
        if (res < 0.0) res = 0.0;
        if (res > 1.0) res = 1.0;
        kq = 4.0 * res * resonance_multiplier * q_adjust;
        drive_scaled = ladder_drive_scale((STREAMS && c.drive) ? (double)c.drive[i] : c.p_drive, c.pbg);
        ac0 = alpha * 0.76923077;
        ac1 = alpha * 0.23076923;
        oma = 1.0 - alpha;
    };
    if (!STREAMS) coefficients(0);
    const int oversample = OS ? OS : c.oversample;               // OS = 2: a constant trip count, unrolled

    const int mode = EMIT ? __builtin_amdgcn_readfirstlane(c.mode) : 0;
    const bool mode_uniform = EMIT ? (bool)__all(c.mode == mode) : true;
    auto sample = [&](float x, int64_t i) -> float {
        if (STREAMS) coefficients(i);
        const double input_sample = (double)x * drive_scaled;
        const double decay = fabs(input_sample) < input_threshold ? state_decay : 1.0;   // a select: no branch
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s.z0[q] *= decay;
            s.z1[q] *= decay;
        }
        s.old_input *= decay;
        double interp = 0.0, total = 0.0;
#pragma unroll
        for (int os = 0; os < oversample; ++os) {
            const double in_interp = __builtin_fma(interp, s.old_input, (1.0 - interp) * input_sample);
            const double g = __builtin_fma(-(s.z1[3] - c.pbg * in_interp), kq, in_interp);
            // a stage, ft = alpha * (c0*in + c1*z0 - z1) + z1, regrouped as (alpha*c0) * in + w with
            // w = (alpha*c1) * z0 + (1 - alpha) * z1 known before the stage input is: one dependent operation per
            // stage on the chain through the four stages instead of two
            double w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = __builtin_fma(ac1, s.z0[q], oma * s.z1[q]);
            const double u = FAST ? ladder_tanh_fast(g) : ladder_tanh_mid(g);
            double stage_in = u, st[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double ft = __builtin_fma(ac0, stage_in, w[q]);
                s.z1[q] = ft;
                s.z0[q] = stage_in;
                stage_in = ft;
                st[q] = ft;
            }
            if (EMIT) {                                          // the output taps of ladder_pe.py:183-197
                const int m = mode_uniform ? mode : c.mode;
                double weighted;
                if (m == 0) weighted = st[3];
                else if (m == 1) weighted = st[1];
                else if (m == 2) weighted = (st[1] + st[3]) * 4.0 - st[2] * 8.0;
                else if (m == 3) weighted = (st[0] - st[1]) * 2.0;
                else if (m == 4) weighted = u + st[3] - (st[0] + st[2]) * 4.0 + st[1] * 6.0;
                else weighted = u + st[1] - st[0] * 2.0;
                total += weighted * c.oversample_recip;
            }
            interp += c.oversample_recip;
        }
        s.old_input = input_sample;
        return (float)total;
    };
    int64_t base = i0;
    if (i1 - i0 >= kLadderChunk) {
        float xn[kLadderChunk];
        ladder_load8(c, base, xn);
        for (; base + kLadderChunk <= i1; base += kLadderChunk) {
            float xc[kLadderChunk], yc[kLadderChunk];
#pragma unroll
            for (int j = 0; j < kLadderChunk; ++j) xc[j] = xn[j];
            if (base + 2 * kLadderChunk <= i1) ladder_load8(c, base + kLadderChunk, xn);
#pragma unroll
            for (int j = 0; j < kLadderChunk; ++j) yc[j] = sample(xc[j], base + j);
            if (EMIT) ladder_store8(c, base, yc);
        }
    }
    for (; base < i1; ++base) {
        const float y = sample(c.x[base * c.channels + c.ch], base);
        if (EMIT) c.o[base * c.channels + c.ch] = y;
    }
}

// A time segment's own samples [i0, i1), written (k_ladder_segments).
__device__ __forceinline__ void ladder_emit(const LadderConsts &c, LadderState &s, int64_t i0, int64_t i1) {
    if (c.freq || c.resonance || c.drive) {                                      // kernel arguments: uniform
        ladder_warm_impl<false, 0, true, true>(c, s, i0, i1);
        return;
    }
    if (__all(c.oversample == 2)) ladder_warm_impl<false, 2, false, true>(c, s, i0, i1);
    else ladder_warm_impl<false, 0, false, true>(c, s, i0, i1);
}

template <bool FAST>
__device__ __forceinline__ void ladder_warm(const LadderConsts &c, LadderState &s, int64_t i0, int64_t i1) {
    if (c.freq || c.resonance || c.drive) {                                      // kernel arguments: uniform
        ladder_warm_impl<FAST, 0, true>(c, s, i0, i1);
        return;
    }
    if (__all(c.oversample == 2)) ladder_warm_impl<FAST, 2>(c, s, i0, i1);      // every active lane: wave-uniform
    else ladder_warm_impl<FAST, 0>(c, s, i0, i1);
}

__device__ __forceinline__ void ladder_advance(const LadderConsts &c, LadderState &s, int64_t i0, int64_t emit_from,
                                               int64_t i1, double *snap = nullptr) {
    const int mode_u = __builtin_amdgcn_readfirstlane(c.mode);
    if (c.freq || c.resonance || c.drive) {                       // kernel arguments: uniform
        ladder_advance_impl<true, 0, false>(c, s, i0, emit_from, i1, snap, 0);
    } else if (__all(c.oversample == 2 && c.mode == mode_u)) {    // every active lane: a wave-uniform choice
        ladder_advance_impl<false, 2, true>(c, s, i0, emit_from, i1, snap, mode_u);
    } else {
        ladder_advance_impl<false, 0, false>(c, s, i0, emit_from, i1, snap, 0);
    }
}

__device__ __forceinline__ LadderState ladder_load(const double *st) {
    LadderState s;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s.z0[j] = st[j];
        s.z1[j] = st[4 + j];
    }
    s.old_input = st[8];
    return s;
}
__device__ __forceinline__ void ladder_store(double *st, const LadderState &s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        st[j] = s.z0[j];
        st[4 + j] = s.z1[j];
    }
    st[8] = s.old_input;
}

__global__ void __launch_bounds__(64)
k_ladder(float *out, int64_t out_stride, const float *in, int64_t in_stride, int batch, int64_t n, int channels,
         double sr, const pgx_ladder_params *params, const float *freq, const float *resonance,
         const float *drive, double *state) {
    const int chain = blockIdx.x * 64 + threadIdx.x;
    if (chain >= batch * channels) return;
    const int inst = chain / channels, ch = chain - inst * channels;
    const LadderConsts c = ladder_consts(params[inst], sr, channels, ch, in + (int64_t)inst * in_stride,
                                         out + (int64_t)inst * out_stride, freq, resonance, drive);
    double *st = state + (int64_t)chain * 9;
    LadderState s = ladder_load(st);
    ladder_advance(c, s, 0, 0, n);
    ladder_store(st, s);
}

// One lane per (chain, segment).  warm[chain][seg] = state on entering the segment's first output
// sample, ends[chain][seg] = state after its last one.
template <bool EXACT_EMIT>
__global__ void __launch_bounds__(256)
k_ladder_segments(float *out, int64_t out_stride, const float *in, int64_t in_stride, int batch, int64_t n,
                  int channels, double sr, const pgx_ladder_params *params, const float *freq,
                  const float *resonance, const float *drive, const double *state, int64_t settle,
                  int64_t accurate, int64_t seg_len, int nseg, double *warm, double *ends) {
    // a latency-bound wave: when a throughput-bound kernel shares the chip (the next block's oscillators, rendered
    // beside this kernel) its instructions go first
    __builtin_amdgcn_s_setprio(3);
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t chains = (int64_t)batch * channels;
    if (t >= chains * nseg) return;
    const int chain = (int)(t / nseg), seg = (int)(t - (int64_t)chain * nseg);
    const int inst = chain / channels, ch = chain - inst * channels;
    const pgx_ladder_params p = params[inst];
    const LadderConsts c = ladder_consts(p, sr, channels, ch, in + (int64_t)inst * in_stride,
                                         out + (int64_t)inst * out_stride, freq, resonance, drive);
    const int64_t sb = (int64_t)seg * seg_len;
    int64_t se = sb + seg_len;
    if (se > n) se = n;
    const int64_t ws = sb - settle;
    LadderState s;
    // Every lane runs the same three phases (a wave that took different code paths would run them one after the
    // other): fast warm-up [w0, w1), accurate warm-up [w1, sb), exact samples [sb, se).  A segment whose warm-up
    // would reach before the block start begins at the block start from the carried state instead (nothing it
    // produces before sb is emitted; k_ladder_finish checks it like any other segment).
    int64_t w0, w1;
    const bool scalar_params = !freq && !resonance && !drive;    // kernel arguments: uniform
    if (ws <= 0) {
        s = ladder_load(state + (int64_t)chain * 9);
        w0 = 0;
        w1 = sb - accurate > 0 ? sb - accurate : 0;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) s.z0[j] = s.z1[j] = 0.0;
        s.old_input = (double)c.x[(ws - 1) * channels + ch] *
                      ladder_drive_scale(drive ? (double)drive[ws - 1] : p.drive, c.pbg);
        w0 = ws;
        w1 = sb - accurate > ws ? sb - accurate : ws;
    }
    (void)scalar_params;             // (control streams take the same three phases: ladder_warm reads them per sample)
    ladder_warm<true>(c, s, w0, w1);
    ladder_warm<false>(c, s, w1, sb);
    if (EXACT_EMIT) {                                            // PGX_LADDER_EXACT_EMIT=1: the reference's operation order
        ladder_advance(c, s, sb, sb, se, warm + ((int64_t)chain * nseg + seg) * 9);
    } else {
        ladder_store(warm + ((int64_t)chain * nseg + seg) * 9, s);   // state on entering the first output sample
        ladder_emit(c, s, sb, se);
    }
    ladder_store(ends + ((int64_t)chain * nseg + seg) * 9, s);
}

// One wave per chain: check every warm-started segment against its left neighbour and REPAIR what disagrees.
// A segment whose entry state (from its warm-up) is not the state its left neighbour ended in is rendered again from
// that true state, sample by sample in the reference's operation order, and so is every segment after it until the
// repaired trajectory meets a segment's own entry state again (the ladder re-synchronised: everything that segment and
// its successors emitted stands).  Until round 4 ONE bad boundary re-rendered the whole chain from the block start:
// right, but a look-ahead window of 2.8 M frames has 48 000 boundaries, and with a warm-up found by trial (resonance at
// and above self-oscillation, a beating input) one of them fails now and then -- 1.5 s of one lane for 60 samples'
// worth of disagreement.  The repair costs what it repairs.  A ladder that runs free beside its input fails every
// boundary and is repaired from the first to the last: the sequential render, as before, bit for bit k_ladder.
// counters: [0] chains with a repair (int32), [1] segmented launches (int32), [2..3] samples repaired (int64): all
// cumulative since the workspace was zeroed (ladder_pe.SettleOptimist reads them).
__global__ void __launch_bounds__(64)
k_ladder_finish(float *out, int64_t out_stride, const float *in, int64_t in_stride, int64_t n, int channels,
                double sr, const pgx_ladder_params *params, const float *freq, const float *resonance,
                const float *drive, double *state, int64_t settle, int64_t seg_len, int nseg,
                const double *warm, const double *ends, int *fallbacks) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    if (lane == 0 && chain == 0 && fallbacks) atomicAdd(fallbacks + 1, 1);
    const double *cw = warm + (int64_t)chain * nseg * 9, *ce = ends + (int64_t)chain * nseg * 9;
    auto differs = [](const double *a, const double *b) {
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const double d = fabs(a[j] - b[j]);
            const double m = fmax(fabs(a[j]), fabs(b[j]));
            if (!(d <= 1e-10 + 1e-8 * m)) bad = true;            // also catches NaN
        }
        return bad;
    };
    const int inst = chain / channels, ch = chain - inst * channels;
    const LadderConsts c = ladder_consts(params[inst], sr, channels, ch, in + (int64_t)inst * in_stride,
                                         out + (int64_t)inst * out_stride, freq, resonance, drive);
    LadderState cur{};                                           // lane 0: the true state while a repair is under way
    int repairing = 0;
    long long repaired = 0;
    for (int base = 1; base < nseg; base += 64) {
        const int seg = base + lane;
        const bool bad = seg < nseg && differs(cw + (int64_t)seg * 9, ce + (int64_t)(seg - 1) * 9);
        const unsigned long long mask = __ballot(bad);
        if (mask == 0ull && !repairing) continue;                // (wave-uniform)
        if (lane == 0) {
            for (int b = 0; b < 64 && base + b < nseg; ++b) {
                const int sg = base + b;
                if (repairing) {
                    double now[9];
                    ladder_store(now, cur);
                    if (!differs(now, cw + (int64_t)sg * 9)) {   // re-synchronised: segment sg's own samples stand
                        repairing = 0;
                        continue;
                    }
                } else if ((mask >> b) & 1ull) {
                    if (sg == 1) {
                        // the very first boundary: segment 0 is rendered again too, from the carried state, so that a
                        // chain that fails throughout (a free-running ladder) is the sequential render from its first
                        // sample on, bit for bit (a segment's own samples are on the fused arithmetic otherwise)
                        cur = ladder_load(state + (int64_t)chain * 9);
                        const int64_t e0 = seg_len < n ? seg_len : n;
                        ladder_advance(c, cur, 0, 0, e0);
                        repaired += e0;
                    } else {
                        cur = ladder_load(ce + (int64_t)(sg - 1) * 9);   // its left neighbour's exit: the true trajectory
                    }
                    repairing = 1;
                } else {
                    continue;
                }
                const int64_t sb = (int64_t)sg * seg_len;
                int64_t se = sb + seg_len;
                if (se > n) se = n;
                ladder_advance(c, cur, sb, sb, se);               // the reference's operation order, samples rewritten
                repaired += se - sb;
            }
        }
        repairing = __shfl(repairing, 0);
    }
    double *st = state + (int64_t)chain * 9;
    if (lane == 0) {
        if (repairing) ladder_store(st, cur);                    // repaired through the last segment
        if (repaired) {
            if (fallbacks) {
                atomicAdd(fallbacks, 1);
                atomicAdd(reinterpret_cast<unsigned long long *>(fallbacks) + 1, (unsigned long long)repaired);
            }
        }
    }
    repairing = __shfl(repairing, 0);
    if (!repairing && lane < 9) st[lane] = ce[(int64_t)(nseg - 1) * 9 + lane];
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

namespace {
struct LadderPlan {
    bool segmented;
    int64_t seg_len;
    int nseg;
};

LadderPlan ladder_plan(int batch, int64_t n, int channels, int64_t settle) {
    LadderPlan p{false, n, 1};
    if (settle <= 0) return p;
    const int64_t chains = (int64_t)batch * channels;
    // A lane's run time ~ settle + seg_len, so short segments are good -- until there are more than two waves per
    // CU: measured on C4 (64 chains x 48 000 frames, settle 1024), ms per block by lane count: 8 Ki 0.66, 16 Ki
    // 0.58, 24 Ki 0.56, 32 Ki 0.555, 48 Ki 0.78, 64 Ki 0.83 (the unrolled sample loop is ~40 KB of code: more waves
    // per CU at different places in it fall out of the instruction cache).  Never shorter than 32 samples.
    // With windows of several blocks (voice_bank.py) the segments are long enough for more lanes to pay: C4 in
    // 8-block windows, ms per block by lane count: 16 Ki 0.1205, 24 Ki 0.0928, 32 Ki 0.0805, 40 Ki 0.0742, 48 Ki 0.0710,
    // 64 Ki 0.0736, 96 Ki 0.089.  Both sets of numbers fit time ~ (0.6 settle + seg_len) x cost(lanes), the warm-up
    // being mostly float32-tanh steps and cost = what a step slows down by as the waves per CU grow: the plan is the
    // lane count that minimises it (PGX_LADDER_LANES fixes the count).
    static const int64_t lanes_fixed = getenv("PGX_LADDER_LANES") ? atoll(getenv("PGX_LADDER_LANES")) : 0;
    int64_t lanes_target = lanes_fixed;
    if (lanes_target <= 0) {
        static const struct { int64_t lanes; double cost; } kChoices[] = {
            {16384, 0.96}, {24576, 0.97}, {32768, 1.0}, {40960, 1.04}, {49152, 1.08}, {65536, 1.27}};
        double best = 0.0;
        for (const auto &choice : kChoices) {
            double seg = (double)n * (double)chains / (double)choice.lanes;
            if (seg < 32.0) seg = 32.0;
            const double t = (0.6 * (double)settle + seg) * choice.cost;
            if (lanes_target <= 0 || t < best) {
                best = t;
                lanes_target = choice.lanes;
            }
        }
    }
    int64_t seg_len = pgx::ceil_div(n * chains, lanes_target);
    if (seg_len < 32) seg_len = 32;
    if (n < 2 * (settle + seg_len)) return p;                     // nothing to win
    p.segmented = true;
    p.seg_len = seg_len;
    p.nseg = (int)pgx::ceil_div(n, seg_len);
    return p;
}
}  // namespace

size_t pgx_ladder_workspace_bytes(int batch, int64_t n, int channels, int64_t settle_frames) {
    if (batch <= 0 || n <= 0 || channels <= 0) return 0;
    const LadderPlan p = ladder_plan(batch, n, channels, settle_frames);
    if (!p.segmented) return 0;
    return ((size_t)batch * channels * p.nseg * 18 + 2) * sizeof(double);      // 16 bytes of counters first
}

int pgx_ladder(float *out, int64_t out_stride, const float *in, int64_t in_stride, int batch, int64_t n,
               int channels, double sample_rate, const pgx_ladder_params *params, const float *freq,
               const float *resonance, const float *drive, double *state, int64_t settle_frames,
               int64_t accurate_frames, void *workspace) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && params && state && channels >= 1 && sample_rate > 0, "pgx_ladder: bad argument");
    PGX_CHECK_ARG(batch == 1 || (out_stride >= n * channels && in_stride >= n * channels),
                  "pgx_ladder: instance stride too small");
    PGX_CHECK_ARG(batch == 1 || (!freq && !resonance && !drive),
                  "pgx_ladder: per-sample control streams require batch == 1");
    const int chains = batch * channels;
    const LadderPlan p = ladder_plan(batch, n, channels, settle_frames);
    if (!p.segmented) {
        hipLaunchKernelGGL(k_ladder, dim3((chains + 63) / 64), dim3(64), 0, pgx::stream(), out, out_stride, in,
                           in_stride, batch, n, channels, sample_rate, params, freq, resonance, drive, state);
        PGX_LAUNCH_CHECK("k_ladder");
        return PGX_OK;
    }
    PGX_CHECK_ARG(workspace != nullptr, "pgx_ladder: workspace required for the segmented path");
    // The workspace begins with the counters (wherever its owner keeps it, whatever the block length: a caller that
    // tries warm-up lengths reads them back, ladder_pe.SettleOptimist): [0] chains re-rendered sequentially because a
    // segment's warm-up had not converged (cumulative since the workspace was zeroed), [1] segmented launches.
    int *fallbacks = (int *)workspace;
    double *warm = (double *)workspace + 2;
    double *ends = warm + (size_t)chains * p.nseg * 9;
    const int64_t lanes = (int64_t)chains * p.nseg;
    // (64-thread workgroups, two per CU, and 256-thread ones -- a wave on every SIMD of half the CUs -- measure the
    // same; 64-thread ones with 64 Ki lanes are a third slower: they are not spread over the SIMDs)
    static const int wg = getenv("PGX_LADDER_WG") ? atoi(getenv("PGX_LADDER_WG")) : 256;
    static const int lds_claim = getenv("PGX_LADDER_LDS") ? atoi(getenv("PGX_LADDER_LDS")) : 160 * 1024;
    const unsigned groups = (unsigned)pgx::ceil_div(lanes, wg);
    // The lanes are dependent float64 chains: a wave wants a SIMD to itself.  Up to 128 workgroups each claim a
    // whole CU's LDS (never touched), so a throughput-bound kernel launched beside this one -- the next block's
    // oscillators (voice_bank.py) -- gets the other CUs instead of sharing these SIMDs' issue slots.
    size_t lds = 0;
    if (wg == 256 && groups <= 128 && lds_claim > 0) {
        lds = (size_t)lds_claim;
        static bool allowed = false;
        if (!allowed && lds > 64 * 1024) {
            PGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ladder_segments<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            PGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ladder_segments<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            allowed = true;
        }
    }
    static const bool exact_emit = getenv("PGX_LADDER_EXACT_EMIT") && atoi(getenv("PGX_LADDER_EXACT_EMIT")) != 0;
    const int64_t tail = accurate_frames > 0 && accurate_frames < settle_frames ? accurate_frames : settle_frames;
    if (exact_emit)
        hipLaunchKernelGGL(k_ladder_segments<true>, dim3(groups), dim3(wg), lds, pgx::stream(), out, out_stride, in,
                           in_stride, batch, n, channels, sample_rate, params, freq, resonance, drive,
                           (const double *)state, settle_frames, tail, p.seg_len, p.nseg, warm, ends);
    else
        hipLaunchKernelGGL(k_ladder_segments<false>, dim3(groups), dim3(wg), lds, pgx::stream(), out, out_stride, in,
                           in_stride, batch, n, channels, sample_rate, params, freq, resonance, drive,
                           (const double *)state, settle_frames, tail, p.seg_len, p.nseg, warm, ends);
    PGX_LAUNCH_CHECK("k_ladder_segments");
    hipLaunchKernelGGL(k_ladder_finish, dim3(chains), dim3(64), 0, pgx::stream(), out, out_stride, in, in_stride, n,
                       channels, sample_rate, params, freq, resonance, drive, state, settle_frames, p.seg_len,
                       p.nseg, (const double *)warm, (const double *)ends, fallbacks);
    PGX_LAUNCH_CHECK("k_ladder_finish");
    return PGX_OK;
}

}  // extern "C"
