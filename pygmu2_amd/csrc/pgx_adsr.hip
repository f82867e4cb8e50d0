// pgx_adsr.hip -- AdsrGatedPE / AdsrTriggeredPE (adsr_pe.py:124-196, :279-335), bit-exact and
// parallel.
//
// The reference is a per-sample Python state machine: emit the level, react to the gate edge /
// trigger, then `env += slope` with clamping.  The float64 accumulation must be reproduced
// exactly -- not just to ~1e-16: whether the attack ends after 480 or 481 steps depends on the
// rounding of the running sum -- so `env_k = env_0 + k*slope` is NOT an acceptable shortcut.
//
// What makes it parallel anyway: while the level stays inside one binade [2^e, 2^(e+1)) every
// addition of the constant slope d rounds to the same grid of spacing u = 2^(e-52).  Writing
// env = E*u and |d| = (D + r)*u with integer E, D and 0 <= r < 1 (all exact power-of-two
// scalings), round-to-nearest gives E' = E +/- Dq, Dq = D + [r > 1/2], as long as no tie
// (r == 1/2) occurs and the sum does not leave the binade.  Inside such a "run" the level is the
// exact progression env_t = env_0 + t*dq (dq = +/-Dq*u, every term representable), so 64 lanes
// emit 64 consecutive samples at once, and "how long does the run last" is answered by comparing
// the candidate levels with the last regular level (one ballot) -- no division anywhere.
// Binade crossings, ties, clamp crossings (>= 1, <= sustain, <= 0), zero levels and gate edges
// take ONE literal reference step.  An ADSR cycle is a few dozen runs.
//
// Two kernels per render:
//   k_adsr_edges  fully parallel over (voice, 64-sample chunk): evaluates / loads the control
//                 stream and reduces it to two 64-bit edge masks per chunk (attack, release);
//   k_adsr_walk   one wave per envelope walks its chunks: scalar mask loads, a 512-sample fast
//                 path (one compare), coalesced float32 stores.

#include "pgx_common.h"

namespace {

constexpr int kIdle = 0, kAttack = 1, kDecay = 2, kSustain = 3, kRelease = 4;
constexpr double kTwo52 = 4503599627370496.0;        // 2^52
constexpr double kTwo53 = 9007199254740992.0;        // 2^53
constexpr int kGroupChunks = 8;                       // fast path granularity: 8 x 64 samples
constexpr int kWideWalkBatch = 128;                   // up to here an envelope gets a whole workgroup (see k_adsr_walk)

// ------------------------------------------------------------------------------------------------
// k_adsr_edges
// MODE 0: gate stream in memory; MODE 1: trigger stream in memory; MODE 2: PeriodicGate evaluated
// in-kernel (periodic_gate.py:63-67 over function_gen_pe.py:157-193) -- the [voices][frames] gate
// buffer is never materialised.
// masks[(voice*nchunks + chunk)*2 + {0,1}] = {attack, release} bit per sample of the chunk;
// last_gate[voice] = gate value of the block's last sample (the next block's "previous gate").
// ------------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ float adsr_control(const float *g, const pgx_gate_params &gp, int64_t start, int64_t idx) {
    if (MODE == 2) {
        const double ph0 = pgx::pgx_mod1((double)(start + idx) * gp.dt);
        const double ph = pgx::pgx_mod1(ph0 + gp.phase);
        return (ph < gp.duty) ? 1.0f : 0.0f;
    }
    return g[idx];
}

constexpr int kEdgeRun = 24;          // consecutive chunks of one voice per wave

template <int MODE>
__global__ void __launch_bounds__(256)
k_adsr_edges(unsigned long long *masks, unsigned long long *group_bits, float *last_gate, const float *ctl,
             int64_t ctl_stride, int batch, int64_t start, int64_t n, int64_t nchunks, int64_t gwords,
             const pgx_gate_params *gates, const double *state) {
    const int lane = threadIdx.x & 63;
    // one wave = kEdgeRun consecutive chunks of one voice: the voice's parameters are loaded once and
    // the gate value at the end of a chunk is the "previous sample" of the next one
    const int64_t runs_per_voice = (nchunks + kEdgeRun - 1) / kEdgeRun;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (int64_t)batch * runs_per_voice) return;
    const int inst = (int)(w / runs_per_voice);
    const int64_t c0 = (w - (int64_t)inst * runs_per_voice) * kEdgeRun;
    const int64_t c1 = (c0 + kEdgeRun < nchunks) ? c0 + kEdgeRun : nchunks;
    const float *g = (MODE == 2) ? nullptr : ctl + (int64_t)inst * ctl_stride;
    pgx_gate_params gp{0.0, 0.0, 0.0};
    if (MODE == 2) gp = gates[inst];
    // A periodic gate whose high and low phases both last 65 samples or more changes at most once inside
    // a 64-sample chunk, so "value before the chunk == value at its last sample" means the chunk has no
    // edge: one evaluation per chunk instead of 64 (the common case by a wide margin).
    const double narrow = gp.duty < 1.0 - gp.duty ? gp.duty : 1.0 - gp.duty;
    const bool sparse = (MODE == 2) && gp.dt > 0.0 && narrow >= 65.0 * gp.dt;

    // gate value just before the run (wave-uniform)
    float before = 0.0f;
    if (MODE != 1) {
        before = (c0 == 0) ? (float)state[(int64_t)inst * 3 + 2]
                           : ((MODE == 2) ? adsr_control<MODE>(g, gp, start, c0 * 64 - 1) : g[c0 * 64 - 1]);
    }
    for (int64_t chunk = c0; chunk < c1; ++chunk) {
        const int64_t wave = (int64_t)inst * nchunks + chunk;
        const int64_t i_first = chunk * 64;
        const int64_t i_last = (i_first + 63 < n - 1) ? i_first + 63 : n - 1;
        if (sparse) {
            const float v_last = adsr_control<MODE>(g, gp, start, i_last);
            if (before == v_last) {
                if (lane == 0) {
                    masks[wave * 2 + 0] = 0ull;
                    masks[wave * 2 + 1] = 0ull;
                    if (i_last == n - 1) last_gate[inst] = v_last;
                }
                continue;
            }
        }
        const int64_t idx = i_first + lane;
        const bool valid = idx < n;
        const float cur = valid ? adsr_control<MODE>(g, gp, start, idx) : 0.0f;
        unsigned long long am, rm = 0ull;
        if (MODE == 1) {
            am = __ballot(valid && cur > 0.0f);                        // adsr_pe.py:297: trigger > 0
        } else {
            // previous sample: lane 0 takes the value carried along the run, the other lanes their left
            // neighbour through a whole-wave DPP shift (wave_shr:1)
            const float pv = __int_as_float(__builtin_amdgcn_update_dpp(
                __float_as_int(before), __float_as_int(cur), 0x138, 0xf, 0xf, false));
            am = __ballot(valid && pv == 0.0f && cur == 1.0f);         // adsr_pe.py:146-147
            rm = __ballot(valid && pv == 1.0f && cur == 0.0f);
            if (idx == n - 1) last_gate[inst] = cur;
            // the chunk's last valid sample becomes `before` of the next chunk
            before = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur), (int)(i_last - i_first)));
        }
        if (lane == 0) {
            masks[wave * 2 + 0] = am;
            masks[wave * 2 + 1] = rm;
            if (am | rm) {                                             // rare: mark the 512-sample group
                const int64_t grp = chunk / kGroupChunks;
                atomicOr(&group_bits[(int64_t)inst * gwords + (grp >> 6)], 1ull << (grp & 63));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_adsr_walk
// ------------------------------------------------------------------------------------------------
struct AdsrCtx {
    int s;
    double env;
    long long ends_at;
    // cached description of the current run (valid while `have`):
    bool have;
    int dir;            // +1: level rises, regular while env <= lim; -1: falls, regular while env >= lim; 0: flat
    double dq;          // exact per-sample increment
    double lim;         // last level from which one more step is still regular
};

// Derive the run parameters from (state, env).  Returns false when the next step must be taken
// literally (tie, binade / clamp crossing, zero level, unusual slope sign ...).
__device__ __forceinline__ bool adsr_derive(AdsrCtx &c, const pgx_adsr_params &p, bool triggered, long long now) {
    c.dq = 0.0;
    c.dir = 0;
    c.lim = 0.0;
    if (c.s == kIdle) return c.env == 0.0;
    if (c.s == kSustain) {
        if (c.env != p.sustain_level) return false;
        return !triggered || now < c.ends_at;
    }
    const double d = (c.s == kAttack) ? p.attack_dvdt : (c.s == kDecay ? p.decay_dvdt : p.release_dvdt);
    if ((c.s == kAttack) ? (d < 0.0) : (d > 0.0)) return false;    // only the ordinary slope signs
    if (d != d) return false;
    const double env = c.env;
    if (!(env >= 1e-290) || !(env < 1e290)) return false;          // zero, negative, tiny, inf, nan
    const int e = (int)((__double_as_longlong(env) >> 52) & 0x7ff) - 1023;     // env positive, normal
    if (c.s == kAttack && e >= 0) return false;                    // env >= 1: the clamp fires next
    const double inv_u = __longlong_as_double((long long)(52 - e + 1023) << 52);   // 2^(52-e)
    const double u = __longlong_as_double((long long)(e - 52 + 1023) << 52);       // 2^(e-52)
    const double q = fabs(d) * inv_u;                              // exact (power-of-two scaling)
    if (!(q < kTwo53)) return false;                               // |d| >= 2^(e+1): leaves the binade
    const double D = floor(q);
    const double r = q - D;
    if (r == 0.5) return false;                                    // tie: round-half-even
    const double Dq = D + (r > 0.5 ? 1.0 : 0.0);
    if (Dq == 0.0) {                                               // |d| < u/2: the level cannot move
        return !(c.s == kDecay && env <= p.sustain_level);
    }
    if (d > 0.0) {
        // from level E the step is regular while E + D + 1 <= 2^53 - 1
        const double top = (kTwo53 - 2.0) - D;
        if (top < kTwo52) return false;
        c.dir = 1;
        c.dq = Dq * u;
        c.lim = top * u;
    } else {
        // regular while E - D - 1 >= 2^52 and the new level stays above the clamp: E - Dq >= F + 1
        const double thr = (c.s == kDecay) ? p.sustain_level : 0.0;
        const double tq = thr * inv_u;
        if (!(tq < kTwo53)) return false;
        double low = kTwo52 + D + 1.0;
        const double low2 = floor(tq) + 1.0 + Dq;
        if (low2 > low) low = low2;
        if (!(low < kTwo53)) return false;
        c.dir = -1;
        c.dq = -(Dq * u);
        c.lim = low * u;
    }
    // the current level itself must allow one regular step
    return (c.dir > 0) ? (env <= c.lim) : (env >= c.lim);
}

// One literal reference step (after the level has been emitted for this sample).
__device__ __forceinline__ void adsr_step(AdsrCtx &c, const pgx_adsr_params &p, bool triggered, long long now) {
    if (c.s == kIdle) {
        c.env = 0.0;
    } else if (c.s == kAttack) {
        c.env += p.attack_dvdt;
        if (c.env >= 1.0) { c.env = 1.0; c.s = kDecay; }
    } else if (c.s == kDecay) {
        c.env += p.decay_dvdt;
        if (c.env <= p.sustain_level) {
            c.env = p.sustain_level;
            if (triggered) c.ends_at = now + p.sustain_samples;
            c.s = kSustain;
        }
    } else if (c.s == kSustain) {
        c.env = p.sustain_level;
        if (triggered && now >= c.ends_at) c.s = kRelease;
    } else {
        c.env += p.release_dvdt;
        if (c.env <= 0.0) { c.env = 0.0; c.s = kIdle; }
    }
    c.have = false;
}

// General path for one chunk of `nvalid` samples: runs, literal steps, edges.  Inlined (a noinline
// call would force the context through scratch memory); its call sites sit in non-unrolled loops so
// there are only two copies of it per kernel.
template <bool TRIG>
__device__ __forceinline__ double adsr_chunk(AdsrCtx &cx, const pgx_adsr_params &p,
                                                       unsigned long long amask, unsigned long long emask,
                                                       int nvalid, long long now0, int lane) {
    AdsrCtx c = cx;
    double mine = 0.0;
    int a = 0;
    while (a < nvalid) {
        if ((emask >> a) & 1ull) {                                 // gate edge / trigger on this sample
            c.s = ((amask >> a) & 1ull) ? kAttack : kRelease;
            c.have = false;
            emask &= ~(1ull << a);
        }
        const unsigned long long later = emask & ~((2ull << a) - 1ull);
        int limit = later ? (__ffsll((long long)later) - 1) : nvalid;
        if (limit > nvalid) limit = nvalid;
        const long long now = now0 + a;
        if (!c.have) c.have = adsr_derive(c, p, TRIG, now);
        if (c.have) {
            // candidate levels of this run on the lanes; a lane is "regular" if one more step from its
            // level is still exact.  The run extends up to the first lane that is not.
            const int t = lane - a;
            const double v = c.env + (double)t * c.dq;             // exact for every lane we will use
            bool reg = (c.dir > 0) ? (v <= c.lim) : ((c.dir < 0) ? (v >= c.lim) : true);
            if (TRIG && c.s == kSustain) reg = (now + t < c.ends_at);
            const unsigned long long bad = __ballot(t >= 0 && t < limit - a && !reg);
            const int take = bad ? (__ffsll((long long)bad) - 1 - a) : (limit - a);
            if (take > 0) {
                if (t >= 0 && t < take) mine = v;
                c.env = c.env + (double)take * c.dq;               // exact: `take` regular steps
                a += take;
                if (bad) {
                    // the sample that ended the run (no edge on it: edges sit at `limit`) takes its literal
                    // step right here instead of costing another trip round the loop
                    if (lane == a) mine = c.env;
                    adsr_step(c, p, TRIG, now0 + a);
                    a += 1;
                }
                continue;
            }
            c.have = false;
        }
        if (lane == a) mine = c.env;                               // literal step for one sample
        adsr_step(c, p, TRIG, now);
        a += 1;
    }
    cx = c;
    return mine;
}

// WPE = waves per envelope.  1: a bank of envelopes, one wave each.  4: a few envelopes (a rank's share of a
// sharded mix): the four waves of a workgroup walk the SAME envelope redundantly -- the walk is scalar control
// flow, identical in each -- and split the emitting (conversions + stores), which is what the fast path spends
// its issue slots on: chunk k of a group is written by wave k mod 4.  No LDS, no barrier.
template <bool TRIG, int WPE>
__global__ void __launch_bounds__(256)
k_adsr_walk(float *out, int64_t out_stride, int batch, int64_t start, int64_t n, int64_t nchunks, int64_t gwords,
            const pgx_adsr_params *params, const unsigned long long *masks, const unsigned long long *group_bits,
            const float *last_gate, double *state) {
    const int lane = threadIdx.x & 63;
    // A dependent chain on one wave: when it shares a SIMD with throughput kernels of a forked block
    // (pgx_adsr_gated_periodic's detach_walk) it must win instruction arbitration, or it is the
    // block's critical path at a third of its speed.
    __builtin_amdgcn_s_setprio(3);
    // readfirstlane makes the indices provably wave-uniform (scalar loads)
    const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int inst = (WPE == 1) ? blockIdx.x * 4 + wave_id : blockIdx.x;
    const int sub = (WPE == 1) ? 0 : wave_id;                    // which share of the chunks this wave writes
    if (inst >= batch) return;
    const pgx_adsr_params p = params[inst];
    float *o = out + (int64_t)inst * out_stride;
    double *st = state + (int64_t)inst * 3;
    const unsigned long long *mk = masks + (int64_t)inst * nchunks * 2;
    const unsigned long long *gb = group_bits + (int64_t)inst * gwords;

    AdsrCtx c;
    c.s = (int)st[0];
    c.env = st[1];
    c.ends_at = TRIG ? (long long)st[2] : 0;
    c.have = false;
    c.dir = 0;
    c.dq = 0.0;
    c.lim = 0.0;
    if (WPE > 1) __syncthreads();           // every wave has read the state before wave 0 may overwrite it

    // The walk is a dependent chain on a single wave, so nothing on the common path may wait for
    // memory: "does group i contain an edge" is one bit of a per-voice bitmap, 64 groups (32 768
    // samples) per word, fetched one word ahead.
    const int64_t full_groups = n / (64 * kGroupChunks);
    unsigned long long word = (full_groups > 0) ? gb[0] : 0ull, word_next = 0ull;
    for (int64_t grp = 0; grp < full_groups; ++grp) {
        if ((grp & 63) == 0) {
            if (grp > 0) word = word_next;
            if (((grp >> 6) + 1) < gwords) word_next = gb[(grp >> 6) + 1];
        }
        const int64_t ch = grp * kGroupChunks;
        const int64_t base = ch * 64;
        const bool has_edge = ((word >> (grp & 63)) & 1ull) != 0ull;
        if (!has_edge) {
            if (!c.have) c.have = adsr_derive(c, p, TRIG, (long long)(start + base));
            if (c.have) {
                // 512 samples stay inside the run iff the 512th level is still regular (monotone)
                const double vlast = c.env + (double)(64 * kGroupChunks - 1) * c.dq;
                bool ok = (c.dir > 0) ? (vlast <= c.lim) : ((c.dir < 0) ? (vlast >= c.lim) : true);
                if (TRIG && c.s == kSustain) ok = ((long long)(start + base) + 64 * kGroupChunks - 1 < c.ends_at);
                if (ok) {
                    const double env = c.env, dq = c.dq;
#pragma unroll
                    for (int k = 0; k < kGroupChunks; ++k)
                        if (WPE == 1 || (k & (WPE - 1)) == sub)
                            o[base + k * 64 + lane] = (float)(env + (double)(k * 64 + lane) * dq);    // exact
                    c.env = env + (double)(64 * kGroupChunks) * dq;
                    continue;
                }
            }
        }
        // general path: fetch the group's 16 masks together (one latency), then walk its chunks
        unsigned long long am[kGroupChunks], rm[kGroupChunks];
#pragma unroll
        for (int k = 0; k < kGroupChunks; ++k) am[k] = rm[k] = 0ull;
        if (has_edge) {                       // groups that are slow only because a run ends have no edges
#pragma unroll
            for (int k = 0; k < kGroupChunks; ++k) {
                am[k] = mk[(ch + k) * 2];
                rm[k] = mk[(ch + k) * 2 + 1];
            }
        }
#pragma unroll 1
        for (int k = 0; k < kGroupChunks; ++k) {
            // pick chunk k's masks with selects (a dynamically indexed register array would go to scratch);
            // a group that is slow only because a run ends in it has none to pick
            unsigned long long amk = 0ull, rmk = 0ull;
            if (has_edge) {
                amk = am[0];
                rmk = rm[0];
#pragma unroll
                for (int j = 1; j < kGroupChunks; ++j) {
                    amk = (k == j) ? am[j] : amk;
                    rmk = (k == j) ? rm[j] : rmk;
                }
            }
            const int64_t i0 = (ch + k) * 64;
            if ((amk | rmk) == 0ull) {
                // no edge in this chunk: the group was slow because a run ends somewhere in it -- most of
                // its chunks still are plain 64-sample pieces of a run
                if (!c.have) c.have = adsr_derive(c, p, TRIG, (long long)(start + i0));
                if (c.have) {
                    const double vlast = c.env + 63.0 * c.dq;
                    bool ok = (c.dir > 0) ? (vlast <= c.lim) : ((c.dir < 0) ? (vlast >= c.lim) : true);
                    if (TRIG && c.s == kSustain) ok = ((long long)(start + i0) + 63 < c.ends_at);
                    if (ok) {
                        if (WPE == 1 || (k & (WPE - 1)) == sub)
                            o[i0 + lane] = (float)(c.env + (double)lane * c.dq);   // exact
                        c.env = c.env + 64.0 * c.dq;
                        continue;
                    }
                }
            }
            const double mine = adsr_chunk<TRIG>(c, p, amk, amk | rmk, 64, (long long)(start + i0), lane);
            if (WPE == 1 || (k & (WPE - 1)) == sub) o[i0 + lane] = (float)mine;
        }
    }
#pragma unroll 1
    for (int64_t ch = full_groups * kGroupChunks; ch < nchunks; ++ch) {     // tail (last chunk may be partial)
        const unsigned long long am = mk[ch * 2], rm = mk[ch * 2 + 1];
        const int64_t i0 = ch * 64;
        const int nvalid = (n - i0 < 64) ? (int)(n - i0) : 64;
        const double mine = adsr_chunk<TRIG>(c, p, am, am | rm, nvalid, (long long)(start + i0), lane);
        if (lane < nvalid && (WPE == 1 || (int)(ch & (WPE - 1)) == sub)) o[i0 + lane] = (float)mine;
    }
    if (lane == 0 && sub == 0) {
        st[0] = (double)c.s;
        st[1] = c.env;
        st[2] = TRIG ? (double)c.ends_at : (double)last_gate[inst];
    }
}

struct AdsrWs {
    unsigned long long *masks;
    unsigned long long *group_bits;
    float *last_gate;
    int64_t nchunks, gwords;
    size_t bits_bytes;
};

AdsrWs adsr_ws(void *workspace, int batch, int64_t n) {
    AdsrWs w;
    w.nchunks = pgx::ceil_div(n, 64);
    w.gwords = pgx::ceil_div(pgx::ceil_div(w.nchunks, kGroupChunks), 64);
    w.masks = (unsigned long long *)workspace;
    w.group_bits = w.masks + (size_t)batch * w.nchunks * 2;
    w.bits_bytes = (size_t)batch * w.gwords * sizeof(unsigned long long);
    w.last_gate = (float *)(w.group_bits + (size_t)batch * w.gwords);
    return w;
}

template <int MODE>
int adsr_launch(float *out, int64_t out_stride, const float *ctl, int64_t ctl_stride, int batch, int64_t start,
                int64_t n, const pgx_gate_params *gates, const pgx_adsr_params *params, double *state,
                void *workspace, bool detach_walk = false) {
    AdsrWs w = adsr_ws(workspace, batch, n);
    PGX_HIP(hipMemsetAsync(w.group_bits, 0, w.bits_bytes, pgx::stream()));
    const int64_t edge_waves = (int64_t)batch * pgx::ceil_div(w.nchunks, kEdgeRun);
    hipLaunchKernelGGL(k_adsr_edges<MODE>, dim3((unsigned)pgx::ceil_div(edge_waves, 4)), dim3(256), 0, pgx::stream(),
                       w.masks, w.group_bits, w.last_gate, ctl, ctl_stride, batch, start, n, w.nchunks, w.gwords,
                       gates, (const double *)state);
    PGX_LAUNCH_CHECK("k_adsr_edges");
    if (detach_walk) {
        // the walk is one latency-bound wave per envelope: it runs behind the edges on the side stream
        // and leaves the main stream (and nearly all of the machine) to the caller until pgx_stream_join()
        int rc = pgx_stream_fork();
        if (rc != PGX_OK) return rc;
    }
    if (batch <= kWideWalkBatch)
        hipLaunchKernelGGL((k_adsr_walk<MODE == 1, 4>), dim3(batch), dim3(256), 0, pgx::stream(), out, out_stride,
                           batch, start, n, w.nchunks, w.gwords, params, (const unsigned long long *)w.masks,
                           (const unsigned long long *)w.group_bits, (const float *)w.last_gate, state);
    else
        hipLaunchKernelGGL((k_adsr_walk<MODE == 1, 1>), dim3((batch + 3) / 4), dim3(256), 0, pgx::stream(), out,
                           out_stride, batch, start, n, w.nchunks, w.gwords, params,
                           (const unsigned long long *)w.masks, (const unsigned long long *)w.group_bits,
                           (const float *)w.last_gate, state);
    PGX_LAUNCH_CHECK("k_adsr_walk");
    if (detach_walk) return pgx_stream_select(0);
    return PGX_OK;
}

}  // namespace

extern "C" {

size_t pgx_adsr_workspace_bytes(int batch, int64_t n) {
    if (batch <= 0 || n <= 0) return 0;
    size_t nchunks = (size_t)pgx::ceil_div(n, 64);
    size_t gwords = (size_t)pgx::ceil_div(pgx::ceil_div((int64_t)nchunks, kGroupChunks), 64);
    return (size_t)batch * (nchunks * 2 + gwords) * sizeof(unsigned long long) + (size_t)batch * sizeof(float) + 64;
}

int pgx_adsr_gated(float *out, int64_t out_stride, const float *gate, int64_t gate_stride, int batch, int64_t n,
                   const pgx_adsr_params *params, double *state, void *workspace) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && gate && params && state && workspace, "pgx_adsr_gated: null pointer");
    PGX_CHECK_ARG(batch == 1 || (out_stride >= n && gate_stride >= n), "pgx_adsr_gated: stride too small");
    return adsr_launch<0>(out, out_stride, gate, gate_stride, batch, 0, n, nullptr, params, state, workspace);
}

int pgx_adsr_gated_periodic(float *out, int64_t out_stride, int batch, int64_t start, int64_t n,
                            const pgx_gate_params *gates, const pgx_adsr_params *params, double *state,
                            void *workspace, int detach_walk) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && gates && params && state && workspace, "pgx_adsr_gated_periodic: null pointer");
    PGX_CHECK_ARG(batch == 1 || out_stride >= n, "pgx_adsr_gated_periodic: stride too small");
    return adsr_launch<2>(out, out_stride, nullptr, 0, batch, start, n, gates, params, state, workspace,
                          detach_walk != 0);
}

int pgx_adsr_triggered(float *out, int64_t out_stride, const float *trig, int64_t trig_stride, int batch,
                       int64_t start, int64_t n, const pgx_adsr_params *params, double *state, void *workspace) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && trig && params && state && workspace, "pgx_adsr_triggered: null pointer");
    PGX_CHECK_ARG(batch == 1 || (out_stride >= n && trig_stride >= n), "pgx_adsr_triggered: stride too small");
    return adsr_launch<1>(out, out_stride, trig, trig_stride, batch, start, n, nullptr, params, state, workspace);
}

}  // extern "C"
