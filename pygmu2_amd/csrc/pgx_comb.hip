// pgx_comb.hip -- CombPE: y[n] = x[n] + fb[n] * y[n - D[n]]  (comb_pe.py:26-113, _comb_process_numba).
//
// Two data paths, chosen by what drives the delay:
//
// (1) scalar frequency (the reference's examples/16_comb_filter.py, every bank): the one-pole smoother sits on its
//     input from the first sample on, so D is ONE integer the host derives with the reference's expression.  A comb
//     with a fixed delay is D independent first-order recurrences (one per residue n mod D): lane j owns the frames
//     j, j + D, j + 2D ... of a time segment, keeps its previous output in a register and never meets another lane --
//     no ring traffic, no barrier.  Long renders are cut into segments of m*D frames that run concurrently:
//     k_comb_poly<REDUCE> leaves every (segment, lane)'s zero-state response Z and feedback product P, and
//     k_comb_poly<APPLY> folds the earlier segments' (P, Z) onto the value the ring holds for the lane and then runs
//     its frames in the reference's operation order (multiply, add; no fused multiply-add).  Up to 1024 steps per
//     lane the render is one segment and one launch: bit for bit the reference's loop.
//     The ring (float64, buffer_len rows) only carries the last buffer_len outputs from one render to the next.  It
//     is double buffered ([2][rows][channels], `parity` names the half to read): lanes that finish early write the
//     new half while late lanes still read their carry-in from the old one.
//
// (2) frequency from a PE: k_comb_delays turns the control stream into integer delays with a time-parallel evaluation
//     of the one-pole (the literal update on each thread's samples, affine composition across threads and, over the
//     4096-sample tiles of a block, across workgroups).  Short blocks: k_comb_ring, one workgroup per channel with the
//     ring in LDS -- chunks of samples that do not reach into themselves (chunk length from a sliding minimum of the
//     delays made by k_comb_delays: no search in the loop) are read, computed and written by the lanes at once, one
//     barrier per chunk; samples, delays and feedback are staged through LDS tile by tile.  Blocks of three 4096-frame
//     segments and more: k_comb_seg_a / _compose / _groups / _b -- every segment at once (comment above k_comb_seg_a).
#include "pgx_common.h"

namespace {

constexpr double kMaxFeedback = 0.995;                  // comb_pe.py:32 (max_feedback passed by _render)
constexpr int kPolySingleSteps = 1024;                  // up to this many steps per lane: one segment (exact)
constexpr int kPolySegSteps = 64;                       // steps per lane and segment when segmented
constexpr int kPolyMaxSeg = 512;                        // the apply pass folds up to this many (P, Z) pairs per lane
constexpr int kPolyBatch = 32;                          // loads in flight per lane (and as many being consumed)

__device__ __forceinline__ double comb_fb(double f) {   // comb_pe.py:87-95
    f = isfinite(f) ? f : 0.0;
    f = f > kMaxFeedback ? kMaxFeedback : f;
    f = f < -kMaxFeedback ? -kMaxFeedback : f;
    return f;
}

struct PolyPlan {
    int64_t steps;      // steps per lane and segment (the last segment may be shorter)
    int nseg;
};
__host__ __device__ inline PolyPlan poly_plan(int64_t n, int64_t d) {
    const int64_t total = (n + d - 1) / d;              // steps of the longest chain
    if (total <= kPolySingleSteps) return PolyPlan{total, 1};
    int64_t m = kPolySegSteps;
    const int64_t need = (total + kPolyMaxSeg - 1) / kPolyMaxSeg;
    if (m < need) m = need;
    return PolyPlan{m, (int)((total + m - 1) / m)};
}

// Workspace of one chain bundle (one voice): [nseg][D*C] pairs (P, Z) as two planes.
__host__ __device__ inline int64_t poly_ws_doubles(int64_t n, int channels, int64_t d_max) {
    // nseg * D <= n / steps + D  <=  n / kPolySegSteps + 2 * D
    return 2 * (int64_t)channels * (n / kPolySegSteps + 2 * d_max + 64);
}

enum { COMB_REDUCE = 0, COMB_APPLY = 1 };

template <int PASS, bool FBS>
__global__ void __launch_bounds__(256)
k_comb_poly(float *out, int64_t out_stride, const float *in, int64_t in_stride, int64_t n, int channels,
            const pgx_comb_params *params, const float *fbs, double *ring, int64_t ring_rows, int64_t total_frames,
            int parity, double *ws, int64_t ws_stride) {
    const int voice = blockIdx.y;
    const pgx_comb_params prm = params[voice];
    const int64_t D = prm.delay, len = prm.buffer_len;
    const int64_t DC = D * channels;
    const PolyPlan plan = poly_plan(n, D);
    const int64_t lanes_total = (int64_t)gridDim.x * 256;
    const int64_t L = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double *P = ws + (int64_t)voice * ws_stride;
    double *Z = P + ws_stride / 2;
    const float *x = in + (int64_t)voice * in_stride;
    const int64_t seg_frames = plan.steps * D;
    const double fbc = comb_fb(prm.feedback);

    if (PASS == COMB_REDUCE) {
        if (L >= (int64_t)(plan.nseg - 1) * DC) return;           // the last segment has no successor
        const int64_t seg = L / DC, chain = L - seg * DC;
        const int64_t r = chain / channels;
        const int64_t e0 = seg * seg_frames * channels + chain;   // first element of the lane
        const int64_t f0 = seg * seg_frames + r;
        double z = 0.0, p = 1.0;
        // A lane's frames are D frames apart: every step is its own memory transaction, and the chain itself is two
        // dependent operations per step.  U loads are kept in flight while the U before them are consumed (a batch
        // fetched, waited for and consumed in turn cost one memory latency per 8 steps: 61 us for a 44 100-frame block).
        constexpr int U = kPolyBatch;
        float xa[U], fa[U];
        // (uniform base pointer + 32-bit lane offset: see APPLY)
        const unsigned xo = (unsigned)e0, fo = (unsigned)f0;
        const unsigned sx = (unsigned)DC, sf = (unsigned)D, last = (unsigned)(plan.steps - 1);
        auto fetch = [&](int64_t k0, float (&xv)[U], float (&fv)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                unsigned k = (unsigned)k0 + u;
                k = k < last ? k : last;
                xv[u] = x[xo + k * sx];
                fv[u] = FBS ? fbs[fo + k * sf] : 0.f;
            }
        };
        auto consume = [&](int64_t k0, const float (&xv)[U], const float (&fv)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (k0 + u < plan.steps) {
                    const double f = FBS ? comb_fb((double)fv[u]) : fbc;
                    z = __builtin_fma(f, z, (double)xv[u]);       // feeds carries only: fused
                    p *= f;
                }
            }
        };
        float xb[U], fb2[U];
        fetch(0, xa, fa);
        for (int64_t k0 = 0; k0 < plan.steps; k0 += 2 * U) {      // two register images used in turn (see APPLY)
            fetch(k0 + U, xb, fb2);
            consume(k0, xa, fa);
            fetch(k0 + 2 * U, xa, fa);
            consume(k0 + U, xb, fb2);
        }
        P[L] = p;
        Z[L] = z;
        return;
    }

    // ---- APPLY
    const double *ring_old = ring + ((int64_t)voice * 2 + parity) * ring_rows * channels;
    double *ring_new = ring + ((int64_t)voice * 2 + (parity ^ 1)) * ring_rows * channels;
    const int64_t wp0 = total_frames % len;
    // rows this render does not overwrite travel to the new half as they are
    if (n < len) {
        for (int64_t idx = L; idx < (len - n) * channels; idx += lanes_total) {
            const int64_t row = (wp0 + n + idx / channels) % len, ch = idx % channels;
            ring_new[row * channels + ch] = ring_old[row * channels + ch];
        }
    }
    if (L >= (int64_t)plan.nseg * DC) return;
    const int64_t seg = L / DC, chain = L - seg * DC;
    const int64_t r = chain / channels, ch = chain - r * channels;
    int64_t row = (wp0 + r - D) % len;
    if (row < 0) row += len;
    double c = ring_old[row * channels + ch];
    {
        // the earlier segments' (P, Z) pairs, a batch of loads at a time
        constexpr int U = 16;
        for (int64_t t0 = 0; t0 < seg; t0 += U) {
            double pv[U], zv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t t = (t0 + u < seg) ? t0 + u : seg - 1;
                pv[u] = P[t * DC + chain];
                zv[u] = Z[t * DC + chain];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (t0 + u < seg) c = __builtin_fma(pv[u], c, zv[u]);
        }
    }

    float *y = out + (int64_t)voice * out_stride;
    const int64_t f0 = seg * seg_frames + r;                      // first frame of the lane
    int64_t steps = plan.steps;
    if (f0 >= n) steps = 0;
    else if (f0 + (steps - 1) * D >= n) steps = (n - 1 - f0) / D + 1;
    const int64_t keep_from = n - len;                            // frames from here on stay in the ring
    if (steps <= 0) return;
    constexpr int U = kPolyBatch;
    // Element offsets inside a voice's block fit 32 bits (the entry point checks n * channels < 2^30): every access is
    // "uniform base pointer + 32-bit lane offset", one address register per access instead of two -- with 2 x 32
    // loads in flight that is the difference between one and two or three waves per SIMD.
    const unsigned xo = (unsigned)(f0 * channels + ch), fo = (unsigned)f0;
    const unsigned sx = (unsigned)DC, sf = (unsigned)D, last = (unsigned)(steps - 1);
    // the lane's first step whose frame stays in the ring (keep_from <= f0 + k * D)
    int64_t keep_k = keep_from <= f0 ? 0 : (keep_from - f0 + D - 1) / D;
    if (keep_k > steps) keep_k = steps;
    auto fetch = [&](unsigned k0, float (&xv)[U], float (&fv)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            unsigned k = k0 + u;
            k = k < last ? k : last;
            xv[u] = x[xo + k * sx];
            fv[u] = FBS ? fbs[fo + k * sf] : 0.f;
        }
    };
    // U whole steps that neither end the lane's run nor reach the ring: nothing but the recurrence and its store
    auto plain = [&](unsigned k0, const float (&xv)[U], const float (&fv)[U]) {
        unsigned off = xo + k0 * sx;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double f = FBS ? comb_fb((double)fv[u]) : fbc;
            const double v = (double)xv[u] + f * c;               // comb_pe.py:97 (multiply, then add)
            y[off] = (float)v;
            off += sx;
            c = v;
        }
    };
    // Phase 1: whole batches before the ring part, two register images used in turn -- the loads of one are in flight
    // while the other is consumed, and nothing waits for them at the end of an iteration (copying one image onto the
    // other did: the wait for the next batch sat behind every batch's arithmetic instead of under it).
    const unsigned whole = (unsigned)(keep_k / U) * U;
    unsigned k = 0;
    if (whole) {
        float xa[U], fa[U], xb[U], fb2[U];
        fetch(0, xa, fa);
        while (k + 2 * U <= whole) {
            fetch(k + U, xb, fb2);
            plain(k, xa, fa);
            fetch(k + 2 * U, xa, fa);                             // (clamped: the last one may fetch past `whole`)
            plain(k + U, xb, fb2);
            k += 2 * U;
        }
        if (k + U <= whole) {
            plain(k, xa, fa);
            k += U;
        }
    }
    // Phase 2: the steps that are left -- the ring part (the last buffer_len frames) and the ragged end
    int64_t slot = (wp0 + f0 + (int64_t)k * D) % len;             // ring row of the lane's current frame
    for (unsigned k0 = k; k0 <= last; k0 += U) {
        float xv[U], fv[U];
        fetch(k0, xv, fv);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned kk = k0 + u;
            if (kk <= last) {
                const double f = FBS ? comb_fb((double)fv[u]) : fbc;
                const double v = (double)xv[u] + f * c;
                y[xo + kk * sx] = (float)v;
                if ((int64_t)kk >= keep_k) ring_new[slot * channels + ch] = v;
                slot += D;
                slot = slot >= len ? slot - len : slot;
                c = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ PE-driven frequency
constexpr int kCtlThreads = 1024, kCtlT = 4, kCtlTile = kCtlThreads * kCtlT;

constexpr int kCtlSegTiles = 1;                               // a workgroup's share of a block: one tile of 4096 samples

// delay[i] = clip(rint(sr / max(sm_i, 1)), 1, len - 1) with sm the one-pole of comb_pe.py:61-68;
// gmin / gmax[g] = min / max of the delays of samples [64 g, 64 g + 64).  state[0] = smoothed frequency (-1: unset).
// One workgroup per segment of kCtlSegTiles tiles.  A block of more than one segment (a look-ahead window) takes two
// launches: PASS 0 runs every segment's one-pole and keeps only where it ends -- segment 0 from the carried level, the
// others from zero (the one-pole is affine in its level: sm_end = F sm_start + Z with one F for all whole segments) --
// and PASS 1 folds those ends onto the carried level to enter its own segment, then produces the delays.  (One
// workgroup walking a 2.8 M-sample window tile by tile took 1.1 ms.)  ends[s] for s < nseg - 1.
template <int PASS>
__global__ void __launch_bounds__(kCtlThreads)
k_comb_delays(int64_t n, double sr, const float *freq, double min_frequency, double alpha, int64_t len, double *state,
              int32_t *delay, int32_t *gmin, int32_t *gmax, double *ends, int nseg) {
    __shared__ double s_wave[kCtlThreads / 64];
    __shared__ double s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = blockIdx.x;
    const int64_t seg_begin = (int64_t)seg * kCtlSegTiles * kCtlTile;
    int64_t seg_end = seg_begin + (int64_t)kCtlSegTiles * kCtlTile;
    if (seg_end > n) seg_end = n;
    if (PASS == 0 && seg == nseg - 1) return;                     // nobody enters after the last segment
    // the homogeneous factor of kCtlT literal steps, its powers along a wave, and the factor of a whole wave
    double at = 1.0;
#pragma unroll
    for (int j = 0; j < kCtlT; ++j) at = at + (0.0 - at) * alpha;
    double pw[6];
    pw[0] = at;
#pragma unroll
    for (int k = 1; k < 6; ++k) pw[k] = pw[k - 1] * pw[k - 1];
    const double aw = pw[5] * pw[5];                              // at^64
    double plane = 1.0;                                           // at^lane
#pragma unroll
    for (int k = 0; k < 6; ++k)
        if (lane & (1 << k)) plane *= pw[k];
    if (tid == 0) {
        double c = state[0];
        if (c < 0.0) {                                            // first sample ever: smoothed = raw (comb_pe.py:65-66)
            c = (double)freq[0];
            c = c < min_frequency ? min_frequency : c;
        }
        // (the last segment of PASS 1 rewrites state[0] when it ends: with several segments the level on entering the
        // block travels through the spare slot ends[nseg - 1], written by PASS 0, not through state[0])
        if (nseg > 1 && seg == 0) {
            if (PASS == 0) ends[nseg - 1] = c;
            else c = ends[nseg - 1];
        }
        if (PASS == 1 && seg == 0) state[1] = c;                  // the level on entering the block: k_comb_delays_literal
        if (seg > 0) {
            if (PASS == 0) {
                c = 0.0;                                          // zero-state response of the segment
            } else {
                // F = (factor of one thread's samples)^(threads x tiles): squarings from at^64
                double F = aw;
#pragma unroll
                for (int t = 64; t < kCtlThreads * kCtlSegTiles; t <<= 1) F = F * F;
                c = ends[0];
                for (int t = 1; t < seg; ++t) c = __builtin_fma(F, c, ends[t]);
            }
        }
        s_carry = c;
    }
    __syncthreads();
    for (int64_t base = seg_begin; base < seg_end; base += kCtlTile) {
        const int64_t i0 = base + (int64_t)tid * kCtlT;
        double raw[kCtlT];
#pragma unroll
        for (int j = 0; j < kCtlT; ++j) {
            const int64_t i = i0 + j < n ? i0 + j : n - 1;
            const double v = (double)freq[i];
            raw[j] = v < min_frequency ? min_frequency : v;
        }
        double z = 0.0;
#pragma unroll
        for (int j = 0; j < kCtlT; ++j) z = z + (raw[j] - z) * alpha;
        double v = z;                                             // inclusive scan along the wave: v_t = z_t + at v_(t-1)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double u = __shfl_up(v, 1 << k);
            if (lane >= (1 << k)) v = __builtin_fma(pw[k], u, v);
        }
        if (lane == 63) s_wave[wave] = v;
        __syncthreads();
        double c = s_carry;
        for (int w = 0; w < wave; ++w) c = __builtin_fma(aw, c, s_wave[w]);
        double e = __shfl_up(v, 1);
        if (lane == 0) e = 0.0;
        double sm = __builtin_fma(plane, c, e);                   // level on entering this thread's samples
        int dmin = 0x7fffffff, dmax = 0;
        bool tie = false;
#pragma unroll
        for (int j = 0; j < kCtlT; ++j) {
            sm = sm + (raw[j] - sm) * alpha;                      // comb_pe.py:68
            if (PASS == 1) {
                const double f = sm < 1.0 ? 1.0 : sm;
                const double q = sr / f;
                // The delay is index work and has to be the reference's integer.  This level and the literal chain's both
                // sit ~1e-14 (relative) from the exact recurrence, so round(q) can differ only where q is within ~1e-10
                // of a half-integer: such a sample is flagged (1e-9: an order of magnitude of margin) and the block's
                // delays are then made again by the literal one-lane chain (k_comb_delays_literal).
                tie = tie || (fabs((q - floor(q)) - 0.5) < 1e-9 && i0 + j < n);
                int64_t d = (int64_t)rint(q);                     // np.round: half to even
                d = d < 1 ? 1 : d;
                d = d >= len ? len - 1 : d;
                if (i0 + j < n) {
                    delay[i0 + j] = (int32_t)d;
                    dmin = min(dmin, (int)d);
                    dmax = max(dmax, (int)d);
                    if (i0 + j == n - 1) state[0] = sm;
                }
            }
        }
        if (PASS == 1) {
            if (__any(tie) && lane == 0) reinterpret_cast<int *>(state + 2)[0] = 1;       // (every writer writes 1)
#pragma unroll
            for (int s = 1; s < 64 / kCtlT; s <<= 1) {
                dmin = min(dmin, __shfl_xor(dmin, s));
                dmax = max(dmax, __shfl_xor(dmax, s));
            }
            if ((tid & (64 / kCtlT - 1)) == 0 && i0 < n) {
                gmin[i0 >> 6] = dmin;
                gmax[i0 >> 6] = dmax;
            }
        }
        __syncthreads();                                          // s_wave / s_carry read by everyone
        if (tid == kCtlThreads - 1) {
            s_carry = sm;
            if (PASS == 0 && base + kCtlTile >= seg_end) ends[seg] = sm;      // a whole segment: its last thread's level
        }
        __syncthreads();
    }
}

// The reference's loop itself (comb_pe.py:61-85), one lane, for a block in which k_comb_delays met a rounding tie's
// neighbourhood: the same operations in the same order give the reference's smoothed frequency bit for bit, hence its
// delays.  Launched after every k_comb_delays<1>; returns at once unless the flag is up (practically always: a sample
// lands within 1e-9 of a tie once in hours of audio).  8 ns per sample when it runs.  state = {smoothed frequency,
// level on entering the block, tie flag (int), blocks redone (int64)}.
__global__ void __launch_bounds__(64)
k_comb_delays_literal(int64_t n, double sr, const float *freq, double min_frequency, double alpha, int64_t len,
                      double *state, int32_t *delay, int32_t *gmin, int32_t *gmax) {
    int *flag = reinterpret_cast<int *>(state + 2);
    if (*flag == 0) return;
    const int lane = threadIdx.x;
    if (lane == 0) {
        double sm = state[1];
        for (int64_t i = 0; i < n; ++i) {
            double raw = (double)freq[i];
            raw = raw < min_frequency ? min_frequency : raw;
            // k_comb_delays entered the block with sm = state[1]; when the stream had never run that value IS the first
            // sample's clamped frequency and the first update leaves it unchanged in the reference too (sm = raw)
            sm = sm + (raw - sm) * alpha;
            const double f = sm < 1.0 ? 1.0 : sm;
            int64_t d = (int64_t)rint(sr / f);
            d = d < 1 ? 1 : d;
            d = d >= len ? len - 1 : d;
            delay[i] = (int32_t)d;
        }
        state[0] = sm;
        reinterpret_cast<long long *>(state + 3)[0] += 1;
    }
    __syncthreads();
    __threadfence();
    for (int64_t g = lane; g < (n + 63) / 64; g += 64) {
        int lo = 0x7fffffff, hi = 0;
        for (int64_t i = g * 64; i < n && i < g * 64 + 64; ++i) {
            const int d = delay[i];
            lo = min(lo, d);
            hi = max(hi, d);
        }
        gmin[g] = lo;
        gmax[g] = hi;
    }
    if (lane == 0) *flag = 0;
}

constexpr int kRingThreads = 256, kRingTile = 2048, kRingGroups = kRingTile / 64, kRingPer = kRingTile / kRingThreads;

// One workgroup per channel.  RING_LDS: the ring lives in LDS (dynamic shared memory, `len` doubles); otherwise in the
// new half of the global ring (min_frequency so low that it does not fit).
template <bool RING_LDS, bool FBS>
__global__ void __launch_bounds__(kRingThreads)
k_comb_ring(float *out, const float *in, int64_t n, int channels, const double *ring_old, double *ring_new,
            int64_t len, int64_t wp0, const int32_t *delay, const int32_t *gmin, const int32_t *gmax,
            const pgx_comb_params *params, const float *fbs) {
    extern __shared__ double s_dyn[];
    __shared__ float s_x[kRingTile];
    __shared__ int32_t s_d[kRingTile];
    __shared__ float s_f[FBS ? kRingTile : 1];
    __shared__ int32_t s_gmin[kRingGroups + 16], s_gmax[kRingGroups + 16], s_chunk[kRingGroups];
    const int tid = threadIdx.x, ch = blockIdx.x;
    double *ring = RING_LDS ? s_dyn : ring_new;
    const int64_t rs = RING_LDS ? 1 : channels;                   // row stride
    const int64_t ro = RING_LDS ? 0 : ch;
    for (int64_t r = tid; r < len; r += kRingThreads) ring[r * rs + ro] = ring_old[r * channels + ch];
    const double fbc = FBS ? 0.0 : comb_fb(params[0].feedback);
    const int64_t groups_total = (n + 63) >> 6;
    int64_t wp = wp0;
    // a tile's samples, delays and feedback travel HBM -> registers while the tile before it is in the chunk loop
    float xr[kRingPer], fr[kRingPer];
    int32_t dr[kRingPer], gl = 0x7fffffff, gh = 0;
    auto fetch = [&](int64_t tb) {
#pragma unroll
        for (int k = 0; k < kRingPer; ++k) {
            int64_t i = tb + tid + k * kRingThreads;
            i = i < n ? i : n - 1;
            xr[k] = in[i * channels + ch];
            dr[k] = delay[i];
            fr[k] = FBS ? fbs[i] : 0.f;
        }
        int64_t g = (tb >> 6) + tid;
        const bool live = tid < kRingGroups + 16 && g < groups_total;
        g = g < groups_total ? g : groups_total - 1;
        const int32_t a = gmin[g], b = gmax[g];
        gl = live ? a : 0x7fffffff;
        gh = live ? b : 0;
    };
    fetch(0);
    for (int64_t tb = 0; tb < n; tb += kRingTile) {
        const int tl = (int)((n - tb) < kRingTile ? (n - tb) : kRingTile);
        __syncthreads();                                          // the previous tile's chunks are done with s_*
#pragma unroll
        for (int k = 0; k < kRingPer; ++k) {
            const int idx = tid + k * kRingThreads;
            s_x[idx] = xr[k];
            s_d[idx] = dr[k];
            if (FBS) s_f[idx] = fr[k];
        }
        if (tid < kRingGroups + 16) {
            s_gmin[tid] = gl;
            s_gmax[tid] = gh;
        }
        __syncthreads();
        if (tb + kRingTile < n) fetch(tb + kRingTile);
        // chunk length for a chunk that starts anywhere in group g: it covers at most kRingThreads samples = the groups
        // g .. g + kRingThreads / 64: no sample may reach into the chunk (S <= min D) nor wrap onto a slot the chunk
        // writes (S <= len - max D)
        if (tid < kRingGroups) {
            int lo = 0x7fffffff, hi = 0;
#pragma unroll
            for (int k = 0; k <= kRingThreads / 64; ++k) {
                lo = min(lo, s_gmin[tid + k]);
                hi = max(hi, s_gmax[tid + k]);
            }
            const int s = min(lo, (int)(len - hi));
            s_chunk[tid] = min(max(s, 1), kRingThreads);
        }
        __syncthreads();
        // The chunk loop is one dependent chain per chunk (ring read -> multiply-add -> ring write -> barrier); all
        // that does not depend on the ring is taken off it: every wave keeps the tile's chunk lengths in one register
        // (lane g = group g: a v_readlane with a scalar index instead of an LDS round trip per chunk), and a chunk's
        // delays / samples / feedback are read from LDS while the chunk before it is still in flight.
        const int my_chunk = s_chunk[(tid & 63) < kRingGroups ? (tid & 63) : 0];
        const int len32 = (int)len;
        int wp32 = (int)wp;
        int p = 0;
        int S = min(__builtin_amdgcn_readlane(my_chunk, 0), tl);
        int ic = tid < tl ? tid : tl - 1;
        int d_cur = s_d[ic];
        float x_cur = s_x[ic], f_cur = FBS ? s_f[ic] : 0.f;
        while (p < tl) {
            const int pn = p + S;
            int Sn = 0;
            if (pn < tl) Sn = min(__builtin_amdgcn_readlane(my_chunk, pn >> 6), tl - pn);
            int in = pn + tid;
            in = in < tl ? in : tl - 1;
            const int d_nxt = s_d[in];
            const float x_nxt = s_x[in], f_nxt = FBS ? s_f[in] : 0.f;
            if (tid < S) {
                int rp = wp32 + tid - d_cur;
                rp = rp < 0 ? rp + len32 : rp;
                const double f = FBS ? comb_fb((double)f_cur) : fbc;
                const double v = (double)x_cur + f * ring[(int64_t)rp * rs + ro];     // comb_pe.py:97
                out[(tb + p + tid) * channels + ch] = (float)v;
                int w = wp32 + tid;
                w = w >= len32 ? w - len32 : w;
                ring[(int64_t)w * rs + ro] = v;
            }
            __syncthreads();
            wp32 += S;
            wp32 = wp32 >= len32 ? wp32 - len32 : wp32;
            p = pn;
            S = Sn;
            d_cur = d_nxt;
            x_cur = x_nxt;
            f_cur = f_nxt;
        }
        wp = wp32;
    }
    if (RING_LDS) {
        __syncthreads();
        for (int64_t r = tid; r < len; r += kRingThreads) ring_new[r * channels + ch] = ring[r];
    }
}

// ------------------------------------------------------------------------------------------------
// PE-driven frequency, long blocks: time segments.  y[n] = x[n] + f[n] y[n - D[n]] has ONE tap, so following the tap
// back from any frame of a segment ends on exactly one frame before the segment: y[n] = Z[n] + P[n] * y[src[n]] with Z
// the frame's zero-state value, P the product of the feedbacks along the way and src the frame the chain lands on
// (all three built chunk by chunk like k_comb_ring builds y: Z[n] = x[n] + f Z[n - D], P[n] = f P[n - D], src[n] =
// src[n - D] while n - D is inside the segment).  Segments are kSegLen frames (>= the longest delay: a chain lands
// in the segment just before), so
//   k_comb_seg_a     every segment at once: (Z, P, src) of its frames, in LDS, left in HBM;
//   the tails        each segment's last `tail` frames (tail = buffer_len - 1 >= any delay) as a function of the tail
//                    before it, composed over groups of segments (k_comb_seg_compose, k_comb_seg_groups below);
//   k_comb_seg_b     every frame at once: y = Z + P * (previous segment's tail)[src], float32 out, new ring.
// The combination y = Z + P * y_prev rounds differently from the loop (<= 1e-7 of peak asserted, ~1e-16 observed in
// float64); blocks of fewer than three segments keep k_comb_ring, which is the loop.
// ------------------------------------------------------------------------------------------------
constexpr int kSegLen = 4096, kSegThreads = 256, kSegGroups = kSegLen / 64;
inline bool comb_stream_segmented(int64_t n, int64_t buffer_len) {
    return buffer_len >= 2 && buffer_len - 1 <= kSegLen - 1 && n >= 3 * (int64_t)kSegLen;
}

template <bool FBS>
__global__ void __launch_bounds__(kSegThreads)
k_comb_seg_a(const float *in, int64_t n, int channels, const int32_t *delay, const int32_t *gmin,
             const pgx_comb_params *params, const float *fbs, double *Zg, double *Pg, int32_t *srcg) {
    extern __shared__ double s_dyn[];                              // Z, P (float64), src, d (int32), x, f (float32): 128 KB
    double *Zs = s_dyn, *Ps = s_dyn + kSegLen;
    int32_t *s_src = reinterpret_cast<int32_t *>(Ps + kSegLen), *s_d = s_src + kSegLen;
    float *s_x = reinterpret_cast<float *>(s_d + kSegLen), *s_f = s_x + kSegLen;
    __shared__ int32_t s_chunk[kSegGroups];
    const int tid = threadIdx.x, ch = blockIdx.y;
    const int64_t seg0 = (int64_t)blockIdx.x * kSegLen;
    const int sl = (int)((n - seg0) < kSegLen ? (n - seg0) : kSegLen);
    const double fbc = FBS ? 0.0 : comb_fb(params[0].feedback);
    const int64_t groups_total = (n + 63) >> 6;
    // the segment's samples, delays and feedback: one coalesced sweep into LDS
#pragma unroll
    for (int k = 0; k < kSegLen / kSegThreads; ++k) {
        const int idx = tid + k * kSegThreads;
        int64_t i = seg0 + idx;
        i = i < n ? i : n - 1;
        s_x[idx] = in[i * channels + ch];
        s_d[idx] = delay[i];
        if (FBS) s_f[idx] = fbs[i];
    }
    __syncthreads();
    // chunk length for a chunk starting in group g (it covers at most kSegThreads frames = groups g .. g + 4): S <= min D
    if (tid < kSegGroups) {
        int lo = 0x7fffffff;
#pragma unroll
        for (int k = 0; k <= kSegThreads / 64; ++k) {
            const int64_t g = (seg0 >> 6) + tid + k;
            if (g < groups_total) lo = min(lo, gmin[g]);
        }
        s_chunk[tid] = min(max(lo, 1), kSegThreads);
    }
    __syncthreads();
    const int my_chunk = s_chunk[(tid & 63) < kSegGroups ? (tid & 63) : 0];
    int p = 0;
    int S = min(__builtin_amdgcn_readlane(my_chunk, 0), sl);
    while (p < sl) {
        const int pn = p + S;
        int Sn = 0;
        if (pn < sl) Sn = min(__builtin_amdgcn_readlane(my_chunk, pn >> 6), sl - pn);
        if (tid < S) {
            const int i = p + tid;
            const int j = i - s_d[i];                              // the frame the tap reads, segment-local
            const double f = FBS ? comb_fb((double)s_f[i]) : fbc;
            double z = (double)s_x[i], pr = f;
            int sr = j;
            if (j >= 0) {
                z = z + f * Zs[j];                                 // comb_pe.py:97 on the zero-state values
                pr = f * Ps[j];
                sr = s_src[j];
            }
            Zs[i] = z;
            Ps[i] = pr;
            s_src[i] = sr;                                         // < 0: frames before the segment's first
        }
        __syncthreads();
        p = pn;
        S = Sn;
    }
    const int64_t base = ((int64_t)ch * ((n + kSegLen - 1) / kSegLen) + blockIdx.x) * kSegLen;
#pragma unroll
    for (int k = 0; k < kSegLen / kSegThreads; ++k) {
        const int idx = tid + k * kSegThreads;
        if (idx < sl) {
            Zg[base + idx] = Zs[idx];
            Pg[base + idx] = Ps[idx];
            srcg[base + idx] = s_src[idx];
        }
    }
}

// The tails.  tail(s) = the last `tail` outputs of segment s is a gather-affine function of tail(s - 1):
// tail(s)[t] = Z[t] + P[t] * tail(s - 1)[tail + src[t]], and two such maps compose into one of the same form
// (Z'' = Z + P Z'[src], P'' = P P'[src], src'' = src'[src]).  Walking 689 segments of a look-ahead window one after the other
// cost 2 us each; instead
//   k_comb_seg_compose  groups of kSegGroupLen segments, all groups at once: segment s's map composed onto its group's
//                       earlier ones -> comp(s): tail(s) as a function of the tail before the group;
//   k_comb_seg_groups   one workgroup per channel walks the GROUPS: T(g + 1) = comp(last of g)(T(g)), T(0) = the ring;
//   k_comb_seg_b        every frame at once: y = Z + P * tail(s - 1)[src], tail(s - 1)[k] = comp(s - 1)(T(group))[k].
constexpr int kSegGroupLen = 32, kFoldThreads = 1024, kFoldPer = 4;          // tail <= 4095 = 4 x 1024 - 1

__global__ void __launch_bounds__(kFoldThreads)
k_comb_seg_compose(int64_t n, int64_t len, const double *Zg, const double *Pg, const int32_t *srcg, double *Zc,
                   double *Pc, int32_t *srcc) {
    extern __shared__ double s_dyn[];                              // two images of (Z, P: float64; src: int32) x tail
    const int tid = threadIdx.x, ch = blockIdx.y;
    const int tail = (int)len - 1;
    const int64_t nseg = (n + kSegLen - 1) / kSegLen;
    const int64_t s_first = (int64_t)blockIdx.x * kSegGroupLen;
    int64_t s_end = s_first + kSegGroupLen;
    if (s_end > nseg - 1) s_end = nseg - 1;                        // the last segment's tail is never needed
    if (s_first >= s_end) return;
    double *zi[2] = {s_dyn, s_dyn + 2 * tail};
    double *pi[2] = {s_dyn + tail, s_dyn + 3 * tail};
    int32_t *si[2] = {reinterpret_cast<int32_t *>(s_dyn + 4 * tail), reinterpret_cast<int32_t *>(s_dyn + 4 * tail) + tail};
    double zr[kFoldPer], pr[kFoldPer];
    int sr[kFoldPer];
    auto fetch = [&](int64_t sgm) {
        const int64_t base = ((int64_t)ch * nseg + sgm) * kSegLen + (kSegLen - tail);
#pragma unroll
        for (int k = 0; k < kFoldPer; ++k) {
            int t = tid + k * kFoldThreads;
            t = t < tail ? t : tail - 1;
            zr[k] = Zg[base + t];
            pr[k] = Pg[base + t];
            sr[k] = srcg[base + t];
        }
    };
    fetch(s_first);
    int img = 0;
    for (int64_t sgm = s_first; sgm < s_end; ++sgm) {
        double zc[kFoldPer], pc[kFoldPer];
        int sc[kFoldPer];
#pragma unroll
        for (int k = 0; k < kFoldPer; ++k) { zc[k] = zr[k]; pc[k] = pr[k]; sc[k] = sr[k]; }
        if (sgm + 1 < s_end) fetch(sgm + 1);                        // the next segment's map, in flight during this one
        const int64_t ob = ((int64_t)ch * nseg + sgm) * tail;
#pragma unroll
        for (int k = 0; k < kFoldPer; ++k) {
            const int t = tid + k * kFoldThreads;
            if (t < tail) {
                double z = zc[k], pp = pc[k];
                int sidx = sc[k];
                if (sgm > s_first) {                                // onto the maps composed so far
                    const int at = tail + sidx;
                    z = __builtin_fma(pp, zi[img ^ 1][at], z);
                    pp = pp * pi[img ^ 1][at];
                    sidx = si[img ^ 1][at];
                }
                zi[img][t] = z;
                pi[img][t] = pp;
                si[img][t] = sidx;
                Zc[ob + t] = z;
                Pc[ob + t] = pp;
                srcc[ob + t] = sidx;
            }
        }
        __syncthreads();
        img ^= 1;
    }
}

// Tg[ch][g][t]: the tail in front of group g (g = 0: the ring).  One workgroup per channel.
__global__ void __launch_bounds__(kFoldThreads)
k_comb_seg_groups(int64_t n, int channels, const double *ring_old, int64_t len, int64_t wp0, const double *Zc,
                  const double *Pc, const int32_t *srcc, double *Tg) {
    extern __shared__ double s_dyn[];                              // two images of `tail` doubles
    const int tid = threadIdx.x, ch = blockIdx.x;
    const int tail = (int)len - 1;
    const int64_t nseg = (n + kSegLen - 1) / kSegLen;
    const int64_t ngroups = (nseg - 1 + kSegGroupLen - 1) / kSegGroupLen;       // groups of segments that have successors
    double *prev = s_dyn, *cur = s_dyn + tail;
    for (int t = tid; t < tail; t += kFoldThreads) {                // frame -tail + t sits in ring row (wp0 - tail + t) mod len
        int64_t row = (wp0 - tail + t) % len;
        row = row < 0 ? row + len : row;
        const double v = ring_old[row * channels + ch];
        prev[t] = v;
        Tg[((int64_t)ch * ngroups + 0) * tail + t] = v;
    }
    __syncthreads();
    for (int64_t g = 0; g + 1 < ngroups; ++g) {
        int64_t last = (g + 1) * kSegGroupLen - 1;                  // the group's last segment (it has successors)
        const int64_t cb = ((int64_t)ch * nseg + last) * tail;
        for (int t = tid; t < tail; t += kFoldThreads) {
            const double y = __builtin_fma(Pc[cb + t], prev[tail + srcc[cb + t]], Zc[cb + t]);
            cur[t] = y;
            Tg[((int64_t)ch * ngroups + g + 1) * tail + t] = y;
        }
        __syncthreads();
        double *sw = prev;
        prev = cur;
        cur = sw;
    }
}

__global__ void __launch_bounds__(256)
k_comb_seg_b(float *out, int64_t n, int channels, const double *ring_old, double *ring_new, int64_t len, int64_t wp0,
             const double *Zg, const double *Pg, const int32_t *srcg, const double *Zc, const double *Pc,
             const int32_t *srcc, const double *Tg) {
    const int ch = blockIdx.y;
    const int tail = (int)len - 1;
    const int64_t nseg = (n + kSegLen - 1) / kSegLen;
    const int64_t ngroups = (nseg - 1 + kSegGroupLen - 1) / kSegGroupLen;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t sgm = i / kSegLen;
    const int local = (int)(i - sgm * kSegLen);
    const int64_t at = ((int64_t)ch * nseg + sgm) * kSegLen + local;
    const int sr = srcg[at];                                        // -tail .. -1
    double yp;
    if (sgm == 0) {
        int64_t row = (wp0 + sr) % len;
        row = row < 0 ? row + len : row;
        yp = ring_old[row * channels + ch];
    } else {
        // tail(sgm - 1)[k] = comp(sgm - 1)(T(group of sgm - 1))[k]
        const int64_t ps = sgm - 1, g = ps / kSegGroupLen;
        const int64_t cb = ((int64_t)ch * nseg + ps) * tail + (tail + sr);
        yp = __builtin_fma(Pc[cb], Tg[((int64_t)ch * ngroups + g) * tail + (tail + srcc[cb])], Zc[cb]);
    }
    const double y = __builtin_fma(Pg[at], yp, Zg[at]);
    out[i * channels + ch] = (float)y;
    if (i >= n - len) ring_new[((wp0 + i) % len) * channels + ch] = y;      // (n >= 3 segments > len: every row is rewritten)
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

size_t pgx_comb_workspace_bytes(int batch, int64_t n, int channels, int delay_max, int freq_stream) {
    if (batch <= 0 || n <= 0 || channels <= 0) return 0;
    if (freq_stream) {     // delays, group minima / maxima, segment ends of the control one-pole; then per frame and
        // channel Z, P (float64), src (int32) and the segment tails of the three-pass path (delay_max = buffer_len here)
        size_t b = (size_t)(n + 2 * ((n + 63) / 64) + 64) * sizeof(int32_t) +
                   (size_t)(pgx::ceil_div(n, (int64_t)kCtlSegTiles * kCtlTile) + 4) * sizeof(double);
        if (comb_stream_segmented(n, delay_max)) {
            const size_t nseg = (size_t)pgx::ceil_div(n, kSegLen);
            b += 16 + (size_t)channels * nseg * ((size_t)kSegLen * 20 + (size_t)delay_max * 20) +
                 (size_t)channels * ((size_t)pgx::ceil_div((int64_t)nseg, kSegGroupLen) + 1) * (size_t)delay_max * 8;
        }
        return b;
    }
    if (delay_max < 1) return 0;
    return (size_t)batch * (size_t)poly_ws_doubles(n, channels, delay_max) * sizeof(double);
}

int pgx_comb(float *out, int64_t out_stride, const float *in, int64_t in_stride, int batch, int64_t n, int channels,
             double sample_rate, const pgx_comb_params *params, int delay_min, int delay_max, const float *freq,
             const float *fb, double min_frequency, int64_t smoothing_samples, double *ring, int64_t ring_rows,
             int64_t total_frames, int parity, double *state, void *workspace) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && ring && channels >= 1 && sample_rate > 0 && ring_rows >= 2 && total_frames >= 0 &&
                      (parity == 0 || parity == 1),
                  "pgx_comb: bad argument");
    PGX_CHECK_ARG(batch == 1 || (out_stride >= n * channels && in_stride >= n * channels),
                  "pgx_comb: voice stride too small");
    PGX_CHECK_ARG(batch == 1 || (!freq && !fb), "pgx_comb: per-sample control streams require batch == 1");
    PGX_CHECK_ARG(n * channels < ((int64_t)1 << 30), "pgx_comb: block too long (n * channels must stay below 2^30)");
    if (freq) {
        // ---- delays from the control stream, then the ring in LDS
        PGX_CHECK_ARG(state && workspace && smoothing_samples >= 1 && min_frequency >= 1.0,
                      "pgx_comb: frequency stream needs state, workspace, smoothing_samples >= 1");
        const int64_t len = ring_rows;
        int32_t *delay = (int32_t *)workspace;
        int32_t *gmin = delay + n, *gmax = gmin + (n + 63) / 64;
        double *ends = (double *)(gmax + (n + 63) / 64 + 2);       // (8-byte aligned: see pgx_comb_workspace_bytes)
        ends = (double *)(((uintptr_t)ends + 7) & ~(uintptr_t)7);
        const int ctl_segs = (int)pgx::ceil_div(n, (int64_t)kCtlSegTiles * kCtlTile);
        if (ctl_segs > 1) {
            hipLaunchKernelGGL(k_comb_delays<0>, dim3(ctl_segs), dim3(kCtlThreads), 0, pgx::stream(), n, sample_rate,
                               freq, min_frequency, 1.0 / (double)smoothing_samples, len, state, delay, gmin, gmax,
                               ends, ctl_segs);
            PGX_LAUNCH_CHECK("k_comb_delays<ends>");
        }
        hipLaunchKernelGGL(k_comb_delays<1>, dim3(ctl_segs), dim3(kCtlThreads), 0, pgx::stream(), n, sample_rate, freq,
                           min_frequency, 1.0 / (double)smoothing_samples, len, state, delay, gmin, gmax, ends,
                           ctl_segs);
        PGX_LAUNCH_CHECK("k_comb_delays");
        hipLaunchKernelGGL(k_comb_delays_literal, dim3(1), dim3(64), 0, pgx::stream(), n, sample_rate, freq,
                           min_frequency, 1.0 / (double)smoothing_samples, len, state, delay, gmin, gmax);
        PGX_LAUNCH_CHECK("k_comb_delays_literal");
        const double *ring_old = ring + (int64_t)parity * ring_rows * channels;
        double *ring_new = ring + (int64_t)(parity ^ 1) * ring_rows * channels;
        const int64_t wp0 = total_frames % len;
        PGX_CHECK_ARG(fb || params, "pgx_comb: params required for a scalar feedback");
        static const bool seg_on = !(getenv("PGX_COMB_SEGMENTS") && atoi(getenv("PGX_COMB_SEGMENTS")) == 0);
        if (seg_on && comb_stream_segmented(n, len)) {
            // ---- three passes over time segments (comment above k_comb_seg_a)
            const int64_t nseg = pgx::ceil_div(n, kSegLen);
            const int tail = (int)len - 1;
            double *Zg = (double *)(((uintptr_t)(ends + ctl_segs + 2) + 7) & ~(uintptr_t)7);
            double *Pg = Zg + (size_t)channels * nseg * kSegLen;
            double *Zc = Pg + (size_t)channels * nseg * kSegLen;
            double *Pc = Zc + (size_t)channels * nseg * tail;
            const int64_t ngroups = pgx::ceil_div(nseg - 1, kSegGroupLen);
            double *Tg = Pc + (size_t)channels * nseg * tail;
            int32_t *srcg = (int32_t *)(Tg + (size_t)channels * ngroups * tail);
            int32_t *srcc = srcg + (size_t)channels * nseg * kSegLen;
            static bool allowed = false;
            if (!allowed) {
                PGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_comb_seg_a<true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, kSegLen * 32));
                PGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_comb_seg_a<false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, kSegLen * 32));
                PGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_comb_seg_compose),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 40));
                PGX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_comb_seg_groups),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4096 * 8));
                allowed = true;
            }
            const dim3 ga((unsigned)nseg, (unsigned)channels);
            if (fb)
                hipLaunchKernelGGL(k_comb_seg_a<true>, ga, dim3(kSegThreads), kSegLen * 32, pgx::stream(), in, n, channels,
                                   (const int32_t *)delay, (const int32_t *)gmin, params, fb, Zg, Pg, srcg);
            else
                hipLaunchKernelGGL(k_comb_seg_a<false>, ga, dim3(kSegThreads), kSegLen * 32, pgx::stream(), in, n, channels,
                                   (const int32_t *)delay, (const int32_t *)gmin, params, fb, Zg, Pg, srcg);
            PGX_LAUNCH_CHECK("k_comb_seg_a");
            hipLaunchKernelGGL(k_comb_seg_compose, dim3((unsigned)ngroups, (unsigned)channels), dim3(kFoldThreads),
                               (size_t)tail * 40, pgx::stream(), n, len, (const double *)Zg, (const double *)Pg,
                               (const int32_t *)srcg, Zc, Pc, srcc);
            PGX_LAUNCH_CHECK("k_comb_seg_compose");
            hipLaunchKernelGGL(k_comb_seg_groups, dim3(channels), dim3(kFoldThreads), (size_t)2 * tail * 8, pgx::stream(), n,
                               channels, ring_old, len, wp0, (const double *)Zc, (const double *)Pc, (const int32_t *)srcc, Tg);
            PGX_LAUNCH_CHECK("k_comb_seg_groups");
            hipLaunchKernelGGL(k_comb_seg_b, dim3((unsigned)pgx::ceil_div(n, 256), (unsigned)channels), dim3(256), 0,
                               pgx::stream(), out, n, channels, ring_old, ring_new, len, wp0, (const double *)Zg,
                               (const double *)Pg, (const int32_t *)srcg, (const double *)Zc, (const double *)Pc,
                               (const int32_t *)srcc, (const double *)Tg);
            PGX_LAUNCH_CHECK("k_comb_seg_b");
            return PGX_OK;
        }
        const size_t lds = (size_t)len * sizeof(double);
        const bool in_lds = lds <= 96 * 1024;
        if (in_lds && lds > 32 * 1024) {
            static bool allowed[2] = {false, false};
            if (!allowed[fb ? 1 : 0]) {
                const void *fn = fb ? reinterpret_cast<const void *>(k_comb_ring<true, true>)
                                    : reinterpret_cast<const void *>(k_comb_ring<true, false>);
                PGX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
                allowed[fb ? 1 : 0] = true;
            }
        }
#define PGX_RING(LDS, FBS)                                                                                        \
    hipLaunchKernelGGL((k_comb_ring<LDS, FBS>), dim3(channels), dim3(kRingThreads), (LDS) ? lds : 0, pgx::stream(), \
                       out, in, n, channels, ring_old, ring_new, len, wp0, (const int32_t *)delay,                \
                       (const int32_t *)gmin, (const int32_t *)gmax, params, fb)
        if (in_lds) {
            if (fb) PGX_RING(true, true);
            else PGX_RING(true, false);
        } else {
            if (fb) PGX_RING(false, true);
            else PGX_RING(false, false);
        }
#undef PGX_RING
        PGX_LAUNCH_CHECK("k_comb_ring");
        return PGX_OK;
    }
    // ---- scalar frequency: polyphase chains
    PGX_CHECK_ARG(params && delay_min >= 1 && delay_max >= delay_min && delay_max < ring_rows,
                  "pgx_comb: bad delay range");
    const PolyPlan worst = poly_plan(n, delay_min);               // most segments
    int64_t lanes = 0;                                            // the widest voice
    {
        const PolyPlan a = poly_plan(n, delay_min), b = poly_plan(n, delay_max);
        const int64_t la = (int64_t)a.nseg * delay_min, lb = (int64_t)b.nseg * delay_max;
        // nseg * D is not monotonic in D; bound it: nseg * D <= n / steps + D
        lanes = (la > lb ? la : lb);
        const int64_t bound = n / kPolySegSteps + 2 * (int64_t)delay_max;
        if (delay_min != delay_max) lanes = bound;
        lanes *= channels;
    }
    const int64_t ws_stride = poly_ws_doubles(n, channels, delay_max);
    const bool segmented = worst.nseg > 1 || poly_plan(n, delay_max).nseg > 1;
    PGX_CHECK_ARG(!segmented || workspace != nullptr, "pgx_comb: workspace required for segmented renders");
    const dim3 grid((unsigned)pgx::ceil_div(lanes, 256), (unsigned)batch);
    if (segmented) {
        if (fb)
            hipLaunchKernelGGL((k_comb_poly<COMB_REDUCE, true>), grid, dim3(256), 0, pgx::stream(), out, out_stride, in,
                               in_stride, n, channels, params, fb, ring, ring_rows, total_frames, parity,
                               (double *)workspace, ws_stride);
        else
            hipLaunchKernelGGL((k_comb_poly<COMB_REDUCE, false>), grid, dim3(256), 0, pgx::stream(), out, out_stride,
                               in, in_stride, n, channels, params, fb, ring, ring_rows, total_frames, parity,
                               (double *)workspace, ws_stride);
        PGX_LAUNCH_CHECK("k_comb_poly<reduce>");
    }
    if (fb)
        hipLaunchKernelGGL((k_comb_poly<COMB_APPLY, true>), grid, dim3(256), 0, pgx::stream(), out, out_stride, in,
                           in_stride, n, channels, params, fb, ring, ring_rows, total_frames, parity,
                           (double *)workspace, ws_stride);
    else
        hipLaunchKernelGGL((k_comb_poly<COMB_APPLY, false>), grid, dim3(256), 0, pgx::stream(), out, out_stride, in,
                           in_stride, n, channels, params, fb, ring, ring_rows, total_frames, parity,
                           (double *)workspace, ws_stride);
    PGX_LAUNCH_CHECK("k_comb_poly<apply>");
    return PGX_OK;
}

}  // extern "C"
