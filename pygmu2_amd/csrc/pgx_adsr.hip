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
// scalings), round-to-nearest gives E' = E +/- Dq, Dq = D + [r > 1/2], as long as the sum does
// not leave the binade; in the one binade where r == 1/2 (a tie on every step) round-half-even
// moves an even E by the even one of D, D + 1.  Inside such a "run" the level is the
// exact progression env_t = env_0 + t*dq (dq = +/-Dq*u, every term representable), so the lanes
// emit consecutive samples at once, and "how long does the run last" is (last regular level -
// level) / dq: integers in units of u, divided in float64 and repaired with an exact remainder.
// Binade crossings, odd levels in a tie binade, clamp crossings (>= 1, <= sustain, <= 0), zero
// levels and gate edges take ONE literal reference step.  An ADSR cycle is a few dozen runs.
//
// Two kernels per render:
//   k_adsr_edges  fully parallel over (voice, 64-sample chunk): evaluates / loads the control
//                 stream and reduces it to two 64-bit edge masks per chunk (attack, release) and a
//                 per-voice bitmap of the 512-sample groups that contain an edge;
//   k_adsr_walk   per envelope, run by run: next edge, derive the run, its length in closed form,
//                 emit it with all lanes (coalesced float32 stores), one literal step where it ends.

#include "pgx_common.h"

namespace {

constexpr int kIdle = 0, kAttack = 1, kDecay = 2, kSustain = 3, kRelease = 4;
constexpr double kTwo52 = 4503599627370496.0;        // 2^52
constexpr double kTwo53 = 9007199254740992.0;        // 2^53
constexpr int kGroupChunks = 8;                       // fast path granularity: 8 x 64 samples
constexpr int kParWaves = 8;                          // k_adsr_walk_par: stretches of a block walked at once
constexpr int kParWalkBatch = 512;                    // up to here the stretches of a block are walked at once (k_adsr_walk_par)
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
             int64_t ctl_stride, int batch, int64_t start_arg, int64_t n, int64_t nchunks, int64_t gwords,
             const pgx_gate_params *gates, const double *state, int64_t start_stride = 0) {
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
    // start_stride != 0: the batch is ONE envelope's consecutive chunks (adsr_run_chunks): instance i is the stretch
    // that starts i * start_stride frames into the render, and all instances share the first gate's parameters
    if (MODE == 2) gp = gates[start_stride ? 0 : inst];
    const int64_t start = start_arg + (int64_t)inst * start_stride;
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

// MODE 2 with a "sparse" PeriodicGate (both phases last >= 65 samples: at most one transition per 64-sample
// chunk): one wave looks at 64 CHUNKS at once -- lane l evaluates the gate at the last sample of chunk c0 + l, its
// left neighbour's value is the chunk's "sample before" -- and only the few chunks where the two differ are
// expanded to their 64 samples.  (k_adsr_edges walks a voice's chunks one after the other, every lane repeating
// the same evaluation: 28 us for 512 voices x 750 chunks; this form: a few us.)  Same masks, same bitmap.  A gate
// that is not sparse has its 64 chunks expanded in turn.
__global__ void __launch_bounds__(256)
k_adsr_edges_sparse(unsigned long long *masks, unsigned long long *group_bits, float *last_gate, int batch,
                    int64_t start_arg, int64_t n, int64_t nchunks, int64_t gwords, const pgx_gate_params *gates,
                    const double *state, int64_t start_stride = 0) {
    const int lane = threadIdx.x & 63;
    const int64_t groups_per_voice = (nchunks + 63) / 64;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (int64_t)batch * groups_per_voice) return;
    const int inst = (int)(w / groups_per_voice);
    const int64_t c0 = (w - (int64_t)inst * groups_per_voice) * 64;
    const pgx_gate_params gp = gates[start_stride ? 0 : inst];      // (start_stride: see k_adsr_edges)
    const int64_t start = start_arg + (int64_t)inst * start_stride;
    // A wave's 64 chunks are 8 groups of kGroupChunks = one BYTE of the voice's group bitmap, which the wave alone
    // writes: a plain store, no atomics, and the bitmap needs no clearing launch in front of this kernel (4.7 us + a
    // dispatch gap on C5's envelope chain).  The voice's last wave clears the padding bytes of the last word.
    static_assert(kGroupChunks == 8, "one byte of the group bitmap per wave");
    unsigned bits = 0;                                             // wave-uniform
    auto store_group_byte = [&](unsigned value) {
        if (lane != 0) return;
        unsigned char *row = (unsigned char *)(group_bits + (int64_t)inst * gwords);
        row[c0 >> 6] = (unsigned char)value;
        if (c0 + 64 >= nchunks)
            for (int64_t k = (c0 >> 6) + 1; k < gwords * 8; ++k) row[k] = 0;
    };
    const double narrow = gp.duty < 1.0 - gp.duty ? gp.duty : 1.0 - gp.duty;
    if (!(gp.dt > 0.0 && narrow >= 65.0 * gp.dt)) {
        // a gate with a phase shorter than 65 samples may flip twice inside a chunk: every chunk of the group is
        // expanded, one after the other, carrying the last sample along (k_adsr_edges' loop)
        float before = (c0 == 0) ? (float)state[(int64_t)inst * 3 + 2] : adsr_control<2>(nullptr, gp, start, c0 * 64 - 1);
        const int64_t c1 = (c0 + 64 < nchunks) ? c0 + 64 : nchunks;
        for (int64_t ch = c0; ch < c1; ++ch) {
            const int64_t idx = ch * 64 + lane;
            const int64_t last = (ch * 64 + 63 < n - 1) ? ch * 64 + 63 : n - 1;
            const bool ok = idx < n;
            const float cur = ok ? adsr_control<2>(nullptr, gp, start, idx) : 0.0f;
            const float pv = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(before), __float_as_int(cur),
                                                                        0x138, 0xf, 0xf, false));
            const unsigned long long am = __ballot(ok && pv == 0.0f && cur == 1.0f);
            const unsigned long long rm = __ballot(ok && pv == 1.0f && cur == 0.0f);
            if (idx == n - 1) last_gate[inst] = cur;
            before = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur), (int)(last - ch * 64)));
            if (lane == 0) {
                masks[((int64_t)inst * nchunks + ch) * 2 + 0] = am;
                masks[((int64_t)inst * nchunks + ch) * 2 + 1] = rm;
            }
            if (am | rm) bits |= 1u << ((ch / kGroupChunks) & 7);
        }
        store_group_byte(bits);
        return;
    }
    const int64_t chunk = c0 + lane;
    const bool valid = chunk < nchunks;
    const int64_t i_first = chunk * 64;
    const int64_t i_last = (i_first + 63 < n - 1) ? i_first + 63 : n - 1;
    const float v_last = valid ? adsr_control<2>(nullptr, gp, start, i_last) : 0.0f;
    float first_before;                                            // the sample before the group (wave-uniform)
    if (c0 == 0) first_before = (float)state[(int64_t)inst * 3 + 2];
    else first_before = adsr_control<2>(nullptr, gp, start, c0 * 64 - 1);
    const float before = __int_as_float(__builtin_amdgcn_update_dpp(
        __float_as_int(first_before), __float_as_int(v_last), 0x138, 0xf, 0xf, false));      // wave_shr:1
    // (the block's first chunk is always expanded: its "sample before" is the carried one, which at the start of a
    // stream -- zero -- or after a seek is not the gate's own previous value, and a rise made of that plus the gate's
    // own fall inside the chunk are two transitions that leave first and last sample equal)
    const bool edge = valid && (before != v_last || chunk == 0);
    if (valid && !edge) {
        masks[((int64_t)inst * nchunks + chunk) * 2 + 0] = 0ull;
        masks[((int64_t)inst * nchunks + chunk) * 2 + 1] = 0ull;
    }
    if (valid && i_last == n - 1) last_gate[inst] = v_last;
    unsigned long long todo = __ballot(edge);
    while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int64_t ch = c0 + l;
        const float bef = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(before), l));
        const int64_t idx = ch * 64 + lane;
        const bool ok = idx < n;
        const float cur = ok ? adsr_control<2>(nullptr, gp, start, idx) : 0.0f;
        const float pv = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(bef), __float_as_int(cur), 0x138,
                                                                    0xf, 0xf, false));
        const unsigned long long am = __ballot(ok && pv == 0.0f && cur == 1.0f);       // adsr_pe.py:146-147
        const unsigned long long rm = __ballot(ok && pv == 1.0f && cur == 0.0f);
        if (lane == 0) {
            masks[((int64_t)inst * nchunks + ch) * 2 + 0] = am;
            masks[((int64_t)inst * nchunks + ch) * 2 + 1] = rm;
        }
        if (am | rm) bits |= 1u << ((ch / kGroupChunks) & 7);
    }
    store_group_byte(bits);
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
    double Dq;
    if (r == 0.5) {
        // Tie binade (the slope's lowest set bit sits exactly half an ulp down: every slope has one such binade,
        // and a release spends e.g. 27 samples in it).  Round-half-even makes this regular too: from an EVEN level
        // E the step lands on the even one of E -+ D, E -+ (D + 1), i.e. moves by D if D is even and by D + 1 if
        // it is odd -- and lands on an even level again.  From an odd level one literal step gets there.
        if (__double2loint(env) & 1) return false;                 // E = env / u is odd (lowest mantissa bit)
        const double half = D * 0.5;
        Dq = (floor(half) == half) ? D : D + 1.0;
    } else {
        Dq = D + (r > 0.5 ? 1.0 : 0.0);
    }
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

// How many consecutive samples, starting with the current one, take a regular step (>= 1 after a successful
// adsr_derive).  Levels and increment are integer multiples of one ulp u with level/u < 2^53, so the quotient is
// taken in float64 and repaired with one exact remainder (fma): no loop over candidates, no ballot.
__device__ __forceinline__ int adsr_run_length(const AdsrCtx &c, bool triggered, long long now, int cap) {
    if (c.dir == 0) {
        if (triggered && c.s == kSustain) return (c.ends_at - now < (long long)cap) ? (int)(c.ends_at - now) : cap;
        return cap;
    }
    const double a = (c.dir > 0) ? (c.lim - c.env) : (c.env - c.lim);      // exact, >= 0
    const double b = fabs(c.dq);
    // the quotient only has to be within one of the truth below `cap` (< 2^31): a Newton-refined reciprocal
    // (relative error ~2^-50) instead of the IEEE division sequence on this dependent chain
    double y = __builtin_amdgcn_rcp(b);
    y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
    double q = floor(a * y);
    if (!(q < 4.0e9)) return cap;                                          // far beyond any cap
    const double r = __builtin_fma(-q, b, a);                              // exact: |r| < 2b
    if (r < 0.0) q -= 1.0;
    else if (r >= b) q += 1.0;
    if (q >= (double)cap) return cap;
    return (int)q + 1;
}

// First sample >= p that carries a gate edge / trigger (or n).  The per-voice bitmap of 512-sample groups that
// contain an edge is searched with bit operations; the eight chunk masks of the group found are fetched by eight
// lanes at once, so an edge costs one memory round trip, not one per chunk on a dependent chain.
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int src_lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffull), src_lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), src_lane);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ int adsr_next_edge(const unsigned long long *mk, const unsigned long long *gb, int nchunks,
                                              int gwords, int n, int p, bool &is_attack, int lane) {
    const int ngroups = (nchunks + kGroupChunks - 1) / kGroupChunks;
    int grp = (p >> 6) / kGroupChunks;
    while (grp < ngroups) {
        // next group at or after `grp` whose bit is set
        int w = grp >> 6;
        unsigned long long word = gb[w] & (~0ull << (grp & 63));
        while (word == 0ull && ++w < gwords) word = gb[w];
        if (word == 0ull) return n;
        grp = w * 64 + (__ffsll((long long)word) - 1);
        const int chunk = grp * kGroupChunks + lane;                       // lanes 0..7: one chunk each
        unsigned long long am = 0ull, m = 0ull;
        if (lane < kGroupChunks && chunk < nchunks) {
            am = mk[(int64_t)chunk * 2];
            m = am | mk[(int64_t)chunk * 2 + 1];
            if (chunk == (p >> 6)) m &= ~0ull << (p & 63);
            else if (chunk < (p >> 6)) m = 0ull;
        }
        const unsigned long long hit = __ballot(m != 0ull);
        if (hit) {
            const int src = __ffsll((long long)hit) - 1;
            const unsigned long long mm = readlane_u64(m, src), aa = readlane_u64(am, src);
            const int bit = __ffsll((long long)mm) - 1;
            is_attack = ((aa >> bit) & 1ull) != 0ull;
            const int at = (grp * kGroupChunks + src) * 64 + bit;
            return at < n ? at : n;
        }
        ++grp;                                                             // its edges all lie before p
    }
    return n;
}

// k_adsr_walk, run by run: between two gate edges the level moves through a few dozen runs (one per binade and
// phase), each an exact arithmetic progression whose length is known in closed form, so the walk is
//   edge? -> derive the run -> its length -> emit `take` samples (all lanes) -> one literal step where it ends.
// WPE = waves per envelope: 1 for a bank of envelopes; 4 for a few (a rank's share of a sharded mix): the four
// waves of a workgroup walk the SAME envelope -- scalar control flow, identical in each -- and split the emitting.
template <bool TRIG, int WPE>
__global__ void __launch_bounds__(256)
k_adsr_walk(float *out, int64_t out_stride, int batch, int64_t start, int64_t n, int64_t nchunks, int64_t gwords,
            const pgx_adsr_params *params, const unsigned long long *masks, const unsigned long long *group_bits,
            const float *last_gate, const double *state, double *state_out) {
    const int lane = threadIdx.x & 63;
    // A dependent chain: when it shares a SIMD with throughput kernels of a forked block
    // (pgx_adsr_gated_periodic's detach_walk) it must win instruction arbitration, or it is the
    // block's critical path at a third of its speed.
    __builtin_amdgcn_s_setprio(3);
    // readfirstlane makes the indices provably wave-uniform (scalar loads)
    const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int inst = (WPE == 1) ? blockIdx.x * 4 + wave_id : blockIdx.x;
    const int sub = (WPE == 1) ? 0 : wave_id;                    // which share of the samples this wave writes
    if (inst >= batch) return;
    const pgx_adsr_params p = params[inst];
    float *o = out + (int64_t)inst * out_stride;
    const double *st = state + (int64_t)inst * 3;
    double *sto = state_out + (int64_t)inst * 3;                 // (the same buffer unless the block is rendered ahead)
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

    // positions inside the block are 32-bit (the entry points refuse longer blocks): 64-bit integer arithmetic and
    // int64 <-> float64 conversions are multi-instruction sequences here, and this loop is instruction-bound
    const int n32 = (int)n, nch32 = (int)nchunks, gw32 = (int)gwords;
    bool edge_attack = false;
    int next_edge = adsr_next_edge(mk, gb, nch32, gw32, n32, 0, edge_attack, lane);
    int pos = 0;
    while (pos < n32) {
        if (pos == next_edge) {                                   // gate edge / trigger on this sample
            c.s = edge_attack ? kAttack : kRelease;
            c.have = false;
            next_edge = adsr_next_edge(mk, gb, nch32, gw32, n32, pos + 1, edge_attack, lane);
        }
        const long long now = (long long)start + pos;
        if (!c.have) c.have = adsr_derive(c, p, TRIG, now);
        if (c.have) {
            const int room = next_edge - pos;                     // >= 1: the run may reach up to the next edge
            const int cnt = adsr_run_length(c, TRIG, now, room + 1);
            const int take = cnt < room ? cnt : room;
            const double env = c.env, dq = c.dq;
            float *dst = o + pos;
            for (int t = sub * 64 + lane; t < take; t += 64 * WPE)
                dst[t] = (float)(env + (double)t * dq);            // exact progression
            c.env = env + (double)take * dq;                       // exact: `take` regular steps
            pos += take;
            if (cnt <= room) {
                // the run ended by itself (not at an edge): its next sample takes a literal step -- unless that
                // sample is an edge, whose state change comes first
                if (pos < n32 && pos != next_edge) {
                    if (sub == 0 && lane == 0) o[pos] = (float)c.env;
                    adsr_step(c, p, TRIG, (long long)start + pos);
                    pos += 1;
                } else {
                    c.have = false;
                }
            }
            continue;
        }
        if (sub == 0 && lane == 0) o[pos] = (float)c.env;          // literal step for one sample
        adsr_step(c, p, TRIG, now);
        pos += 1;
    }
    if (lane == 0 && sub == 0) {
        sto[0] = (double)c.s;
        sto[1] = c.env;
        sto[2] = TRIG ? (double)c.ends_at : (double)last_gate[inst];
    }
}

// ------------------------------------------------------------------------------------------------
// k_adsr_walk_par: several waves per gated envelope, each walking its own stretch of the block.
//
// Between two gate edges the level does not wander for ever: an attack reaches 1, decays to the sustain
// level and STAYS there (exactly `sustain_level`: the clamp assigns it), a release reaches exactly 0 and
// idles.  So if the stretch before an edge is at least as long as that takes from ANY level -- a bound of
// span / |slope| + 8 steps per phase, far above what the rounding of the running sum can add -- the state
// the edge finds is known without walking there: (sustain, S) after an attack edge, (idle, 0) after a
// release edge.  Such an edge is an independent start.  Every wave of the workgroup lists the block's edges
// the same way (one memory round trip: lane l reads the masks of the 512-sample groups l and l + 64), picks
// the independent starts, and wave j walks from start j to start j + 1 exactly as k_adsr_walk does (same
// run arithmetic, same literal steps): the samples are the sequential walk's, bit for bit, whatever the
// partition.  Edges that are not independent starts stay inside the stretch of the wave before them.  C5's
// envelopes (2 Hz gate, 10 / 100 / 200 ms) have four to five independent starts per 48 000-frame block.
// A group holding more than one edge (a gate faster than ~90 Hz) sends the whole envelope down the
// sequential walk in wave 0 -- k_adsr_walk's edge search, masks fetched from memory.
// ------------------------------------------------------------------------------------------------
constexpr long long kNeverSettles = 1ll << 40;
constexpr int kParMaxGroups = 128;                      // 65 536 frames

__device__ __forceinline__ long long adsr_settle_steps(double span, double rate) {
    if (!(rate > 0.0) || !(span >= 0.0)) return kNeverSettles;
    const double q = span / rate;
    if (!(q < 1e12)) return kNeverSettles;
    return (long long)q + 8;
}

#ifndef PGX_ADSR_DEBUG
#define PGX_ADSR_DEBUG(slot, value)        /* tools/microbench/adsr_par_debug.hip records per-envelope facts */
#define PGX_ADSR_CLOCK() 0ll
#define PGX_ADSR_DEBUG_MAX(slot, value)
#endif
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_adsr_walk_par(float *out, int64_t out_stride, int batch, int64_t start, int64_t n, int64_t nchunks, int64_t gwords,
                const pgx_adsr_params *params, const unsigned long long *masks,
                const unsigned long long *group_bits, const float *last_gate, const double *state,
                double *state_out, int shared_params = 0, const int *skip_if_zero = nullptr) {
    if (skip_if_zero && *skip_if_zero == 0) return;              // (adsr_run_chunks: a round nobody needs any more)
    __shared__ int e_pos[WAVES][kParMaxGroups + 2];
    __shared__ unsigned char e_att[WAVES][kParMaxGroups + 2];
    const int lane = threadIdx.x & 63;
    // (a walk a block ahead of the stream -- state_out is another buffer -- is nobody's critical path: it does not
    // take instruction arbitration away from the oscillators it runs beside)
    // (raised for the on-chip mix too, whose voices wait for exactly this walk: the walk 79 -> 73 us, the voices beside it
    // 70 -> 80, the block 103 -> 108 us)
    if (state_out == state) __builtin_amdgcn_s_setprio(3);
    const int j = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int inst = blockIdx.x;
    const pgx_adsr_params p = params[shared_params ? 0 : inst];  // (the batch is one envelope's chunks: adsr_run_chunks)
    float *o = out + (int64_t)inst * out_stride;
    const double *st = state + (int64_t)inst * 3;
    double *sto = state_out + (int64_t)inst * 3;
    const unsigned long long *mk = masks + (int64_t)inst * nchunks * 2;
    const unsigned long long *gb = group_bits + (int64_t)inst * gwords;
    const int s0 = (int)st[0];
    const double env0 = st[1];
    __syncthreads();                         // every wave has read the carried state before one overwrites it

    const int n32 = (int)n, nch32 = (int)nchunks, gw32 = (int)gwords;
    const int ngroups = (nch32 + kGroupChunks - 1) / kGroupChunks;

    // ---- the block's edges, one round trip ----
    int pos_e[2] = {0, 0};
    bool has_e[2] = {false, false}, att_e[2] = {false, false};
    bool crowded = false;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int g = lane + 64 * h;
        if (g < ngroups && ((gb[g >> 6] >> (g & 63)) & 1ull)) {
            int count = 0;
#pragma unroll
            for (int c = 0; c < kGroupChunks; ++c) {
                const int chunk = g * kGroupChunks + c;
                if (chunk < nch32) {
                    const unsigned long long am = mk[(int64_t)chunk * 2], rm = mk[(int64_t)chunk * 2 + 1];
                    const unsigned long long m = am | rm;
                    if (m) {
                        count += __popcll(m);
                        const int bit = __ffsll((long long)m) - 1;
                        pos_e[h] = chunk * 64 + bit;
                        att_e[h] = ((am >> bit) & 1ull) != 0ull;
                    }
                }
            }
            has_e[h] = count == 1 && pos_e[h] < n32;
            crowded = crowded || count > 1;
        }
    }
    const bool sequential = __ballot(crowded) != 0ull;
    const unsigned long long m0 = __ballot(has_e[0]), m1 = __ballot(has_e[1]);
    const int ne0 = __popcll(m0), ne = ne0 + __popcll(m1);
    int *my_pos = e_pos[j];
    unsigned char *my_att = e_att[j];
    const unsigned long long below = (1ull << lane) - 1ull;
    if (has_e[0]) { const int k = __popcll(m0 & below); my_pos[k] = pos_e[0]; my_att[k] = att_e[0] ? 1 : 0; }
    if (has_e[1]) { const int k = ne0 + __popcll(m1 & below); my_pos[k] = pos_e[1]; my_att[k] = att_e[1] ? 1 : 0; }
    if (lane == 0) { my_pos[ne] = n32; my_att[ne] = 0; }
    __builtin_amdgcn_s_waitcnt(0);           // (wave-private LDS rows: the wave reads back what it wrote)
    __builtin_amdgcn_wave_barrier();

    // ---- starts: where a wave may begin with a state it knows ----
    // (a) pinned levels (round 2): an edge that comes after a stretch long enough for the level to be pinned -- exactly S
    //     after an attack edge, exactly 0 after a release edge -- finds a known state: exact, no check needed.
    // (b) attack completions (round 3): an attack always ends by pinning the level to exactly 1.0 (adsr_pe.py:160-163),
    //     so the sample after it starts from (DECAY, 1.0) whatever came before -- what is not known is WHEN: K attack
    //     steps after the edge, K from the level the edge found.  K is an integer that a real-arithmetic model of the
    //     envelope gets right unless (1 - level) / slope sits within ~1e-10 of an integer, and a wrong K of one cycle
    //     rarely changes the next one's (the level at the next attack edge moves by one decay step, 3 % of an attack
    //     step for C5's envelopes).  So every attack edge whose gate stays on long enough for the attack to finish from
    //     level 0 contributes a SPECULATIVE start at edge + K_guess, and the guesses are verified before anything is
    //     written: the wave before each speculative start walks silently up to the attack completion it finds and
    //     publishes where that was; a chain of equalities from wave 0 (the carried state: exact) proves every start, a
    //     mismatch corrects the first wrong position and the round repeats (at most WAVES rounds, then wave 0 walks the
    //     block alone).  Voices gated faster than ~4.5 Hz, which have no pinned start (the decay does not settle inside
    //     the gate's on-time) and were one chain of ~180 runs per block, become one chain per gate cycle.
    __shared__ int s_true[WAVES + 1];
    // the start list (the same in every wave; each wave keeps its own copy in LDS: cheap dynamic indexing)
    __shared__ int l_pos[WAVES][WAVES + 1], l_state[WAVES][WAVES + 1], l_edge[WAVES][WAVES + 1], l_watch[WAVES][WAVES + 1];
    __shared__ double l_env[WAVES][WAVES + 1];
    const double S = p.sustain_level;
    const bool tame = S >= 0.0 && S <= 1.0 && p.attack_dvdt > 0.0 && p.decay_dvdt <= 0.0 && p.release_dvdt < 0.0 &&
                      p.attack_dvdt == p.attack_dvdt && p.decay_dvdt == p.decay_dvdt && p.release_dvdt == p.release_dvdt;
    const long long bD = tame ? adsr_settle_steps(1.0 - S, -p.decay_dvdt) : kNeverSettles;
    auto attack_from = [&](double level) -> long long {
        const long long a = adsr_settle_steps(level < 1.0 ? 1.0 - level : 0.0, p.attack_dvdt);
        return (a < kNeverSettles && bD < kNeverSettles) ? a + bD + 4 : kNeverSettles;
    };
    auto release_from = [&](double level) -> long long { return adsr_settle_steps(level, -p.release_dvdt) + 2; };
    const double top = S > 1.0 ? S : 1.0;                 // no level ever exceeds it

    // the walk of [from, to): k_adsr_walk's loop over this wave's edge list.  emit = false: nothing is written, and the
    // walk stops right after the attack that edge `watch` starts has pinned the level (returns that position + 1 -- the
    // first sample of the decay -- or -1 if it reaches `to` first).
    auto walk = [&](AdsrCtx &c, int from, int to, int next_k, bool emit, int watch) -> int {
        const int *my_pos = e_pos[j];
        const unsigned char *my_att = e_att[j];
        bool edge_attack = false;
        int next_edge = __builtin_amdgcn_readfirstlane(my_pos[next_k]);
        edge_attack = __builtin_amdgcn_readfirstlane((int)my_att[next_k]) != 0;
        if (next_edge > to) next_edge = to;
        int pos = from;
        c.have = false;
        c.dir = 0;
        c.dq = 0.0;
        c.lim = 0.0;
        while (pos < to) {
            if (pos == next_edge) {                                   // gate edge on this sample
                c.s = edge_attack ? kAttack : kRelease;
                c.have = false;
                ++next_k;
                next_edge = __builtin_amdgcn_readfirstlane(my_pos[next_k]);      // my_pos[ne] = n
                edge_attack = __builtin_amdgcn_readfirstlane((int)my_att[next_k]) != 0;
                if (next_edge > to) next_edge = to;
            }
            const long long now = (long long)start + pos;
            if (!c.have) c.have = adsr_derive(c, p, false, now);
            if (c.have) {
                const int room = next_edge - pos;                     // >= 1
                const int cnt = adsr_run_length(c, false, now, room + 1);
                const int take = cnt < room ? cnt : room;
                const double env = c.env, dq = c.dq;
#ifndef PGX_ADSR_NO_EMIT            /* experiments/README.md: what the walk costs without its stores (wrong output) */
                if (emit) {
                    float *dst = o + pos;
                    for (int t = lane; t < take; t += 64) dst[t] = (float)(env + (double)t * dq);
                }
#endif
                c.env = env + (double)take * dq;
                pos += take;
                if (cnt <= room) {
                    if (pos < to && pos != next_edge) {
                        if (emit && lane == 0) o[pos] = (float)c.env;
                        const int before = c.s;
                        adsr_step(c, p, false, (long long)start + pos);
                        pos += 1;
                        if (!emit && before == kAttack && c.s == kDecay && next_k > watch) return pos;
                    } else {
                        c.have = false;
                    }
                }
                continue;
            }
            if (emit && lane == 0) o[pos] = (float)c.env;              // literal step for one sample
            const int before = c.s;
            adsr_step(c, p, false, now);
            pos += 1;
            if (!emit && before == kAttack && c.s == kDecay && next_k > watch) return pos;
        }
        return -1;
    };

    AdsrCtx c;
    c.s = s0;
    c.env = env0;
    c.ends_at = 0;
    if (sequential) {                                     // a crowded group: wave 0 walks the block with k_adsr_walk's
        if (j != 0) return;                               // edge search (masks fetched from memory)
        c.have = false;
        c.dir = 0;
        c.dq = 0.0;
        c.lim = 0.0;
        bool edge_attack = false;
        int next_edge = adsr_next_edge(mk, gb, nch32, gw32, n32, 0, edge_attack, lane);
        int pos = 0;
        while (pos < n32) {
            if (pos == next_edge) {
                c.s = edge_attack ? kAttack : kRelease;
                c.have = false;
                next_edge = adsr_next_edge(mk, gb, nch32, gw32, n32, pos + 1, edge_attack, lane);
            }
            const long long now = (long long)start + pos;
            if (!c.have) c.have = adsr_derive(c, p, false, now);
            if (c.have) {
                const int room = next_edge - pos;
                const int cnt = adsr_run_length(c, false, now, room + 1);
                const int take = cnt < room ? cnt : room;
                const double env = c.env, dq = c.dq;
                float *dst = o + pos;
                for (int t = lane; t < take; t += 64) dst[t] = (float)(env + (double)t * dq);
                c.env = env + (double)take * dq;
                pos += take;
                if (cnt <= room) {
                    if (pos < n32 && pos != next_edge) {
                        if (lane == 0) o[pos] = (float)c.env;
                        adsr_step(c, p, false, (long long)start + pos);
                        pos += 1;
                    } else {
                        c.have = false;
                    }
                }
                continue;
            }
            if (lane == 0) o[pos] = (float)c.env;
            adsr_step(c, p, false, now);
            pos += 1;
        }
        if (lane == 0) {
            sto[0] = (double)c.s;
            sto[1] = c.env;
            sto[2] = (double)last_gate[inst];
        }
        return;
    }

    const long long t_list = PGX_ADSR_CLOCK();
    // ---- the start list: every wave derives the same one from the same edges ----
    int *sp_pos = l_pos[j], *sp_state = l_state[j], *sp_edge = l_edge[j], *sp_watch = l_watch[j];
    double *sp_env = l_env[j];
    int ns = 1;
    if (lane == 0) { sp_pos[0] = 0; sp_state[0] = s0; sp_env[0] = env0; sp_edge[0] = 0; sp_watch[0] = -1; }
    {
        const bool sane0 = env0 >= 0.0 && env0 <= top;
        bool prev_up = s0 == kAttack || s0 == kDecay || s0 == kSustain;
        long long need = !sane0 ? kNeverSettles
                         : s0 == kAttack ? attack_from(env0)
                         : s0 == kDecay ? (tame ? adsr_settle_steps(env0 > S ? env0 - S : 0.0, -p.decay_dvdt) + 2 : kNeverSettles)
                         : s0 == kRelease ? release_from(env0) : 2;
        // a carried state that is pinned already (idle at 0, sustaining at S) makes an edge on the block's very first
        // samples a pinned start like any other
        if (sane0 && ((s0 == kIdle && env0 == 0.0) || (s0 == kSustain && env0 == S))) need = 0;
        // the real-arithmetic model: (ms, ml) at the beginning of sample mp.  Step counts by multiplication with the
        // slopes' reciprocals (a division per phase and edge was most of the 12 us this list took at first): a count
        // that is one off at an exact boundary is a wrong guess like any other -- the verification below decides.
        int ms = s0, mp = 0;
        double ml = env0;
        const bool model_ok = tame && sane0 && p.attack_dvdt > 1e-9;
        const double a_up = p.attack_dvdt, d_dn = -p.decay_dvdt, r_dn = -p.release_dvdt;
        const double inv_a = model_ok ? 1.0 / a_up : 0.0, inv_d = (model_ok && d_dn > 0.0) ? 1.0 / d_dn : 0.0,
                     inv_r = model_ok ? 1.0 / r_dn : 0.0;
        const int k_worst = model_ok ? (int)inv_a + 3 : 0x3fffffff;
        const long long a_full = model_ok ? (long long)inv_a + 8 : kNeverSettles;          // attack from 0, decay from 1:
        const long long d_full = (model_ok && d_dn > 0.0) ? (long long)((1.0 - S) * inv_d) + 8 : (d_dn > 0.0 ? kNeverSettles : 8);
        auto steps_to = [&](double span, double inv) -> double {    // ceil(span / rate), at least 1
            double kk = ceil(span * inv);
            return kk < 1.0 ? 1.0 : kk;
        };
        auto model_advance = [&](int steps) {
            while (steps > 0) {
                if (ms == kAttack) {
                    const double kk = steps_to(1.0 - ml, inv_a);
                    if ((double)steps >= kk) { ml = 1.0; ms = kDecay; steps -= (int)kk; }
                    else { ml += a_up * steps; steps = 0; }
                } else if (ms == kDecay) {
                    if (!(d_dn > 0.0)) { steps = 0; break; }
                    const double kk = steps_to(ml - S, inv_d);
                    if ((double)steps >= kk) { ml = S; ms = kSustain; steps -= (int)kk; }
                    else { ml -= d_dn * steps; steps = 0; }
                } else if (ms == kRelease) {
                    const double kk = steps_to(ml, inv_r);
                    if ((double)steps >= kk) { ml = 0.0; ms = kIdle; steps -= (int)kk; }
                    else { ml -= r_dn * steps; steps = 0; }
                } else {
                    steps = 0;
                }
            }
        };
        (void)a_full; (void)d_full;
        const long long need_att_0 = attack_from(0.0), need_att_s = attack_from(S), need_rel_s = release_from(S),
                        need_rel_0 = release_from(0.0), need_rel_top = release_from(top);
        int prev_pos = 0, prev_start = 0;                       // prev_start: position of the last start listed
        const int *my_pos_l = e_pos[j];
        const unsigned char *my_att_l = e_att[j];
        int nxt = __builtin_amdgcn_readfirstlane(my_pos_l[0]);
        for (int k = 0; k < ne; ++k) {
            const int pos = nxt;
            const bool att = __builtin_amdgcn_readfirstlane((int)my_att_l[k]) != 0;
            nxt = __builtin_amdgcn_readfirstlane(my_pos_l[k + 1]);                // my_pos[ne] = n
            const bool settled = (long long)(pos - prev_pos) >= need;
            if (settled && ns < WAVES) {                          // (a) pinned level at this edge
                if (lane == 0) {
                    sp_pos[ns] = pos; sp_state[ns] = prev_up ? kSustain : kIdle; sp_env[ns] = prev_up ? S : 0.0;
                    sp_edge[ns] = k; sp_watch[ns] = -1;
                }
                ++ns;
                prev_start = pos;
            }
            if (model_ok) {
                // (a settled edge pins the model as well: nothing to advance -- envelopes whose edges all settle, the
                // slow gates, never run the model)
                if (settled) { ms = prev_up ? kSustain : kIdle; ml = prev_up ? S : 0.0; }
                else model_advance(pos - mp);
                mp = pos;
                ms = att ? kAttack : kRelease;
                // (an attack edge that found a pinned level already starts a wave of its own; from exactly 0 the guess
                // would also sit on the boundary -- 1 / slope attack steps in real arithmetic, one more in float64 when
                // attack_time * sr is an integer -- and cost a verification round for nothing)
                if (att && !settled && ns < WAVES && nxt - pos >= k_worst && ml < 1.0) {      // (b) attack completion
                    const int ta = pos + (int)steps_to(1.0 - ml, inv_a);
                    if (ta < n32 && ta > prev_start) {
                        if (lane == 0) {
                            sp_pos[ns] = ta; sp_state[ns] = kDecay; sp_env[ns] = 1.0;
                            sp_edge[ns] = k + 1; sp_watch[ns] = k;
                        }
                        ++ns;
                        prev_start = ta;
                    }
                }
            }
            // the level this edge starts from: pinned if the stretch before it settled, else anything up to `top`
            // (three possible levels per edge kind: their step bounds are made once, outside this loop)
            const bool from_s = settled && prev_up, from_0 = settled ? !prev_up : att;
            prev_pos = pos;
            prev_up = att;
            need = att ? (from_s ? need_att_s : need_att_0) : (from_s ? need_rel_s : from_0 ? need_rel_0 : need_rel_top);
        }
    }
    if (lane == 0) { sp_pos[ns] = n32; sp_watch[ns] = -1; }
    __builtin_amdgcn_s_waitcnt(0);           // (wave-private LDS rows: the wave reads back what it wrote)
    __builtin_amdgcn_wave_barrier();

    const long long t_verify = PGX_ADSR_CLOCK();
    PGX_ADSR_DEBUG_MAX(5, t_verify - t_list);
    // ---- verify the speculative positions (nothing is written yet) ----
    auto L_int = [&](const int *row, int q) { return __builtin_amdgcn_readfirstlane(row[q]); };
    bool any_spec = false;
    for (int q = 1; q < ns; ++q) any_spec = any_spec || L_int(sp_watch, q) >= 0;
    bool give_up = false;
    if (any_spec) {
        for (int round = 0; round < WAVES + 1; ++round) {
            PGX_ADSR_DEBUG(4, round + 1);
            int found = -2;                                       // -2: this wave has nothing to check
            if (j + 1 < ns && L_int(sp_watch, j + 1) >= 0) {      // the start after this wave's is speculative
                AdsrCtx t;
                t.s = L_int(sp_state, j);
                t.env = sp_env[j];
                t.ends_at = 0;
                // (a wrong guess can put the true completion beyond the next start's position: walk on to the block's end)
                found = walk(t, L_int(sp_pos, j), n32, L_int(sp_edge, j), false, L_int(sp_watch, j + 1));
            }
            if (lane == 0) s_true[j + 1] = found;
            __syncthreads();
            // the chain of equalities, the same in every wave
            int first_bad = -1, fix = -1;
            bool ok_prev = true;
            for (int q = 1; q < ns; ++q) {
                bool ok = true;
                if (L_int(sp_watch, q) >= 0) {
                    const int tr = s_true[q];
                    ok = ok_prev && tr == L_int(sp_pos, q);
                    if (!ok && first_bad < 0) { first_bad = q; fix = ok_prev ? tr : -1; }
                }
                ok_prev = ok;
            }
            __syncthreads();                                      // s_true is rewritten in the next round
            if (first_bad < 0) break;
            if (fix < 0 || round == WAVES) { give_up = true; break; }
            if (fix <= L_int(sp_pos, first_bad - 1) || fix >= L_int(sp_pos, first_bad + 1)) { give_up = true; break; }
            if (lane == 0) sp_pos[first_bad] = fix;               // (every wave corrects its own copy)
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (give_up) {                                                // (never seen; keeps the result exact whatever the model did)
        ns = 1;
        if (lane == 0) sp_pos[1] = n32;
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
    }
    PGX_ADSR_DEBUG(0, ns);
    PGX_ADSR_DEBUG(1, give_up ? 1 : 0);
    PGX_ADSR_DEBUG(2, ne);
    PGX_ADSR_DEBUG(3, any_spec ? 1 : 0);

    const long long t_emit = PGX_ADSR_CLOCK();
    PGX_ADSR_DEBUG_MAX(6, t_emit - t_verify);
    // ---- the walk of this wave's stretch, written out ----
    if (j >= ns) return;
    const int seg_start = L_int(sp_pos, j), seg_end = L_int(sp_pos, j + 1);
    c.s = L_int(sp_state, j);
    c.env = sp_env[j];
    walk(c, seg_start, seg_end, L_int(sp_edge, j), true, -1);
    PGX_ADSR_DEBUG_MAX(7, PGX_ADSR_CLOCK() - t_emit);
    if (seg_end == n32 && lane == 0) {                            // the wave that walked to the block's end
        sto[0] = (double)c.s;
        sto[1] = c.env;
        sto[2] = (double)last_gate[inst];
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
                void *workspace, bool detach_walk = false, double *state_out = nullptr, int64_t chunk_stride = 0,
                const int *skip_if_zero = nullptr) {
    // chunk_stride != 0 (adsr_run_chunks): the `batch` instances are consecutive chunks of ONE envelope -- instance i
    // starts i * chunk_stride frames into the render, all share gates[0] / params[0]; walk_par only
    if (state_out == nullptr) state_out = state;                // in place
    AdsrWs w = adsr_ws(workspace, batch, n);
    if (MODE != 2)                                              // (k_adsr_edges_sparse writes every byte of the bitmap itself)
        if (int rc = pgx_memset(w.group_bits, 0, w.bits_bytes)) return rc;
    if (MODE == 2) {
        const int64_t edge_waves = (int64_t)batch * pgx::ceil_div(w.nchunks, 64);
        hipLaunchKernelGGL(k_adsr_edges_sparse, dim3((unsigned)pgx::ceil_div(edge_waves, 4)), dim3(256), 0,
                           pgx::stream(), w.masks, w.group_bits, w.last_gate, batch, start, n, w.nchunks, w.gwords,
                           gates, (const double *)state, chunk_stride);
        PGX_LAUNCH_CHECK("k_adsr_edges_sparse");
    } else {
        const int64_t edge_waves = (int64_t)batch * pgx::ceil_div(w.nchunks, kEdgeRun);
        hipLaunchKernelGGL(k_adsr_edges<MODE>, dim3((unsigned)pgx::ceil_div(edge_waves, 4)), dim3(256), 0,
                           pgx::stream(), w.masks, w.group_bits, w.last_gate, ctl, ctl_stride, batch, start, n,
                           w.nchunks, w.gwords, gates, (const double *)state, chunk_stride);
        PGX_LAUNCH_CHECK("k_adsr_edges");
    }
    if (detach_walk) {
        // the walk is one latency-bound wave per envelope: it runs behind the edges on the side stream
        // and leaves the main stream (and nearly all of the machine) to the caller until pgx_stream_join()
        int rc = pgx_stream_fork();
        if (rc != PGX_OK) return rc;
    }
    static const bool par_on = !(getenv("PGX_ADSR_PAR") && atoi(getenv("PGX_ADSR_PAR")) == 0);
    // (512 envelopes: the walk 101 -> 62 us.  While C5's oscillator + filter kernel took 115 us next to it that only
    // slowed the block down, 0.177 -> 0.186 ms -- the walk was hidden and eight times the waves competed for the SIMDs;
    // with that kernel at 80 us the walk is the critical path: 0.172 -> 0.142 ms.  256 envelopes -- a rank's share at two
    // ranks -- 146 -> 122 us; a rank's 64: walk 100 -> 60 us)
    static const int par_max_batch = getenv("PGX_ADSR_PAR_MAX_BATCH") ? atoi(getenv("PGX_ADSR_PAR_MAX_BATCH")) : kParWalkBatch;
    if (MODE != 1 && (par_on || chunk_stride) && batch <= par_max_batch &&
        pgx::ceil_div(w.nchunks, kGroupChunks) <= kParMaxGroups)
    {
        // sixteen waves per envelope for small banks (a rank's share of a sharded mix): a 65 536-frame chunk of C5's
        // envelopes has 7 - 11 starts, with eight waves some walk two stretches one after the other while most of the chip
        // idles -- 64 envelopes in windows 39.7 -> 36.4 us per block; 128 envelopes: no gain.  PGX_ADSR_PAR_WAVES16_UPTO
        static const int many = getenv("PGX_ADSR_PAR_WAVES16_UPTO") ? atoi(getenv("PGX_ADSR_PAR_WAVES16_UPTO")) : 64;
        if (batch <= many && !chunk_stride)
            hipLaunchKernelGGL(k_adsr_walk_par<16>, dim3(batch), dim3(16 * 64), 0, pgx::stream(), out,
                               out_stride, batch, start, n, w.nchunks, w.gwords, params,
                               (const unsigned long long *)w.masks, (const unsigned long long *)w.group_bits,
                               (const float *)w.last_gate, (const double *)state, state_out, chunk_stride ? 1 : 0,
                               skip_if_zero);
        else
        hipLaunchKernelGGL(k_adsr_walk_par<kParWaves>, dim3(batch), dim3(kParWaves * 64), 0, pgx::stream(), out,
                           out_stride, batch, start, n, w.nchunks, w.gwords, params,
                           (const unsigned long long *)w.masks, (const unsigned long long *)w.group_bits,
                           (const float *)w.last_gate, (const double *)state, state_out, chunk_stride ? 1 : 0,
                           skip_if_zero);
    }
    else if (batch <= kWideWalkBatch)
        hipLaunchKernelGGL((k_adsr_walk<MODE == 1, 4>), dim3(batch), dim3(256), 0, pgx::stream(), out, out_stride,
                           batch, start, n, w.nchunks, w.gwords, params, (const unsigned long long *)w.masks,
                           (const unsigned long long *)w.group_bits, (const float *)w.last_gate, (const double *)state,
                           state_out);
    else
        hipLaunchKernelGGL((k_adsr_walk<MODE == 1, 1>), dim3((batch + 3) / 4), dim3(256), 0, pgx::stream(), out,
                           out_stride, batch, start, n, w.nchunks, w.gwords, params,
                           (const unsigned long long *)w.masks, (const unsigned long long *)w.group_bits,
                           (const float *)w.last_gate, (const double *)state, state_out);
    PGX_LAUNCH_CHECK("k_adsr_walk");
    if (detach_walk) return pgx_stream_select(0);
    return PGX_OK;
}

// Blocks longer than the 65 536 frames k_adsr_walk_par covers (look-ahead windows of a lone envelope: up to 64 x 44 100
// frames) are walked chunk after chunk by it -- the result does not depend on the partition -- instead of in one piece by
// k_adsr_walk's single chain: a 2.8 M-frame window 1.6 ms -> 43 chunks.  PGX_ADSR_CHUNK=0: one piece.
// ---- a lone envelope over a window of many chunks: all chunks at once (round 4) ----
// Walked chunk after chunk a 64-block look-ahead window of a lone AdsrGatedPE is 43 x (edge search + walk) one behind the
// other: 1.4 ms, and the stand-alone envelope rows sat at 10 - 14x the CPU however long the window.  But an envelope
// forgets: every completed attack pins (DECAY, 1.0), so the state a chunk ENDS in does not depend on the state it began in
// as soon as it holds one completed attack.  The chunks are therefore walked as a batch -- instance c = chunk c, the same
// kernels -- in rounds: round 0 starts every chunk from the carried state (wrong for all but chunk 0, yet its exit states
// are right wherever a chunk holds a pin), round 1 starts chunk c from chunk c - 1's exit of round 0.  If the exits of
// round 1 equal those of round 0 bit for bit the entries used were the true ones (a fixed point of the chain is the
// chain: entry 0 is exact, entry c + 1 is chunk c's exit from entry c) and every sample written in round 1 is the
// sequential walk's; else a third round, then the verdict is read on the host (one 4-byte read-back per window) and a
// render that has not settled takes the chunk-after-chunk loop from the untouched carried state.  entries[r][c]: (C + 1) x 3
// doubles per round parity, entries[.][0] = the carried state, a round's walk writes its exits at entries[next] + 3.
__global__ void __launch_bounds__(64)
k_adsr_chunk_entries(double *a0, double *a1, const double *carried, int chunks) {
    for (int i = threadIdx.x; i < (chunks + 1) * 3; i += 64) a0[i] = carried[i % 3];
    if (threadIdx.x < 3) a1[threadIdx.x] = carried[threadIdx.x];
}

// flag = 1 when the exits of two rounds differ anywhere (bitwise), else 0
__global__ void __launch_bounds__(64)
k_adsr_chunk_verify(const double *x, const double *y, int chunks, int *flag) {
    bool differ = false;
    for (int i = threadIdx.x; i < chunks * 3; i += 64)
        differ = differ || __double_as_longlong(x[3 + i]) != __double_as_longlong(y[3 + i]);
    const bool any = __any(differ);
    if (threadIdx.x == 0) *flag = any ? 1 : 0;
}

__global__ void __launch_bounds__(64)
k_adsr_chunk_commit(double *state_out, const double *exits_last, const int *flag) {
    if (*flag == 0 && threadIdx.x < 3) state_out[threadIdx.x] = exits_last[threadIdx.x];
}

struct AdsrChunkWs {
    double *a0, *a1;
    int *flag;
};

size_t adsr_chunk_ws_bytes(int64_t chunks, int64_t chunk_frames) {
    const int64_t nchunks = pgx::ceil_div(chunk_frames, 64);
    const int64_t gwords = pgx::ceil_div(pgx::ceil_div(nchunks, kGroupChunks), 64);
    return (size_t)chunks * (nchunks * 2 + gwords) * sizeof(unsigned long long) + (size_t)chunks * sizeof(float) + 64 +
           (size_t)2 * (chunks + 1) * 3 * sizeof(double) + 64;
}

AdsrChunkWs adsr_chunk_ws(void *workspace, int64_t chunks, int64_t chunk_frames) {
    const AdsrWs w = adsr_ws(workspace, (int)chunks, chunk_frames);
    uintptr_t p = (uintptr_t)(w.last_gate + chunks);
    p = (p + 15) & ~(uintptr_t)15;
    AdsrChunkWs c;
    c.a0 = (double *)p;
    c.a1 = c.a0 + (chunks + 1) * 3;
    c.flag = (int *)(c.a1 + (chunks + 1) * 3);
    return c;
}

template <int MODE>
int adsr_run_chunks(float *out, const float *ctl, int64_t start, int64_t n, const pgx_gate_params *gates,
                    const pgx_adsr_params *params, double *state, void *workspace, double *state_out, bool *settled) {
    constexpr int64_t kParFrames = (int64_t)kParMaxGroups * kGroupChunks * 64;
    const int64_t chunks = n / kParFrames;                     // (full chunks; the caller walks what is left)
    const AdsrChunkWs c = adsr_chunk_ws(workspace, chunks, kParFrames);
    hipLaunchKernelGGL(k_adsr_chunk_entries, dim3(1), dim3(64), 0, pgx::stream(), c.a0, c.a1, (const double *)state,
                       (int)chunks);
    PGX_LAUNCH_CHECK("k_adsr_chunk_entries");
    double *in = c.a0, *nxt = c.a1;
    for (int round = 0; round < 3; ++round) {
        // (round 2 runs only if the exits of rounds 0 and 1 differ: its kernels look at the flag first)
        if (int rc = adsr_launch<MODE>(out, kParFrames, ctl, kParFrames, (int)chunks, start, kParFrames, gates, params, in,
                                       workspace, false, nxt + 3, kParFrames, round == 2 ? c.flag : nullptr))
            return rc;
        if (round >= 1) {
            // exits of this round (in nxt) against the previous round's (in `in`); after round 2 a second verdict
            if (round == 1) {
                hipLaunchKernelGGL(k_adsr_chunk_verify, dim3(1), dim3(64), 0, pgx::stream(), (const double *)nxt,
                                   (const double *)in, (int)chunks, c.flag);
                PGX_LAUNCH_CHECK("k_adsr_chunk_verify");
            }
        }
        double *t = in;
        in = nxt;
        nxt = t;
    }
    // after the loop: `in` holds round 2's exits if it ran, else it was skipped and `nxt` (round 1's) are the last valid
    // ones.  Host verdict: round 1 settled (flag 0) -> exits of round 1; else compare rounds 2 and 1 on the host side
    // of the same read-back.
    int flag_host = 1;
    PGX_HIP(hipMemcpyAsync(&flag_host, c.flag, sizeof(int), hipMemcpyDeviceToHost, pgx::stream()));
    PGX_HIP(hipStreamSynchronize(pgx::stream()));
    double *exits = nxt;                                       // round 1's exits (the buffer round 2 read its entries from)
    if (flag_host != 0) {
        // round 2 ran: settled if its exits equal round 1's
        hipLaunchKernelGGL(k_adsr_chunk_verify, dim3(1), dim3(64), 0, pgx::stream(), (const double *)in,
                           (const double *)nxt, (int)chunks, c.flag);
        PGX_LAUNCH_CHECK("k_adsr_chunk_verify");
        PGX_HIP(hipMemcpyAsync(&flag_host, c.flag, sizeof(int), hipMemcpyDeviceToHost, pgx::stream()));
        PGX_HIP(hipStreamSynchronize(pgx::stream()));
        exits = in;
    }
    *settled = flag_host == 0;
    if (*settled) PGX_HIP(hipMemcpyAsync(state_out, exits + chunks * 3, 3 * sizeof(double), hipMemcpyDeviceToDevice,
                                         pgx::stream()));
    return PGX_OK;
}

// Blocks longer than the 65 536 frames k_adsr_walk_par covers (look-ahead windows of a lone envelope: up to 256 x 44 100
// frames) are walked chunk by chunk by it -- the result does not depend on the partition -- instead of in one piece by
// k_adsr_walk's single chain: all chunks at once for a lone envelope (adsr_run_chunks, which waits for its verdict: one
// device wait per window), else one after the other.  PGX_ADSR_CHUNK=0: one piece; PGX_ADSR_CHUNK_PAR=0: one after the other.
template <int MODE>
int adsr_run(float *out, int64_t out_stride, const float *ctl, int64_t ctl_stride, int batch, int64_t start, int64_t n,
             const pgx_gate_params *gates, const pgx_adsr_params *params, double *state, void *workspace,
             bool detach_walk = false, double *state_out = nullptr) {
    constexpr int64_t kParFrames = (int64_t)kParMaxGroups * kGroupChunks * 64;
    static const bool chunk_on = !(getenv("PGX_ADSR_CHUNK") && atoi(getenv("PGX_ADSR_CHUNK")) == 0);
    static const bool chunk_par = !(getenv("PGX_ADSR_CHUNK_PAR") && atoi(getenv("PGX_ADSR_CHUNK_PAR")) == 0);
    if (MODE == 1 || !chunk_on || batch > kParWalkBatch || n <= kParFrames)
        return adsr_launch<MODE>(out, out_stride, ctl, ctl_stride, batch, start, n, gates, params, state, workspace,
                                 detach_walk, state_out);
    double *carried = state_out ? state_out : state;
    if (MODE != 1 && chunk_par && batch == 1 && !detach_walk && n >= 3 * kParFrames) {
        bool settled = false;
        if (int rc = adsr_run_chunks<MODE>(out, ctl, start, n, gates, params, state, workspace, carried, &settled)) return rc;
        if (settled) {
            const int64_t done = (n / kParFrames) * kParFrames;
            if (done == n) return PGX_OK;
            return adsr_launch<MODE>(out + done, out_stride, ctl ? ctl + done : nullptr, ctl_stride, 1, start + done,
                                     n - done, gates, params, carried, workspace, false, carried);
        }
        // (not settled after three rounds: chunks without a completed attack in a row -- the loop below, from `state`)
    }
    if (detach_walk) {
        // the caller joins: every chunk's edge search and walk go to the side stream (the edge search of a chunk needs the
        // state its predecessor's walk leaves, so it cannot stay on the main stream as it does for a single piece)
        if (int rc = pgx_stream_fork()) return rc;
    }
    for (int64_t pos = 0; pos < n; pos += kParFrames) {
        const int64_t len = n - pos < kParFrames ? n - pos : kParFrames;
        if (int rc = adsr_launch<MODE>(out + pos, out_stride, ctl ? ctl + pos : nullptr, ctl_stride, batch, start + pos, len,
                                       gates, params, pos == 0 ? state : carried, workspace, false, carried))
            return rc;
    }
    if (detach_walk) return pgx_stream_select(0);
    return PGX_OK;
}

}  // namespace

extern "C" {

size_t pgx_adsr_workspace_bytes(int batch, int64_t n) {
    if (batch <= 0 || n <= 0) return 0;
    size_t nchunks = (size_t)pgx::ceil_div(n, 64);
    size_t gwords = (size_t)pgx::ceil_div(pgx::ceil_div((int64_t)nchunks, kGroupChunks), 64);
    size_t bytes = (size_t)batch * (nchunks * 2 + gwords) * sizeof(unsigned long long) + (size_t)batch * sizeof(float) + 64;
    constexpr int64_t kParFrames = (int64_t)kParMaxGroups * kGroupChunks * 64;
    if (batch == 1 && n >= 3 * kParFrames) {                    // a lone envelope's chunks as a batch (adsr_run_chunks)
        const size_t alt = adsr_chunk_ws_bytes(n / kParFrames, kParFrames);
        if (alt > bytes) bytes = alt;
    }
    return bytes;
}

int pgx_adsr_gated(float *out, int64_t out_stride, const float *gate, int64_t gate_stride, int batch, int64_t n,
                   const pgx_adsr_params *params, double *state, void *workspace) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && gate && params && state && workspace, "pgx_adsr_gated: null pointer");
    PGX_CHECK_ARG(n < (int64_t)1 << 30, "pgx_adsr_gated: block too long");
    PGX_CHECK_ARG(batch == 1 || (out_stride >= n && gate_stride >= n), "pgx_adsr_gated: stride too small");
    return adsr_run<0>(out, out_stride, gate, gate_stride, batch, 0, n, nullptr, params, state, workspace);
}

int pgx_adsr_gated_periodic(float *out, int64_t out_stride, int batch, int64_t start, int64_t n,
                            const pgx_gate_params *gates, const pgx_adsr_params *params, double *state,
                            void *workspace, int detach_walk) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && gates && params && state && workspace, "pgx_adsr_gated_periodic: null pointer");
    PGX_CHECK_ARG(n < (int64_t)1 << 30, "pgx_adsr_gated_periodic: block too long");
    PGX_CHECK_ARG(batch == 1 || out_stride >= n, "pgx_adsr_gated_periodic: stride too small");
    return adsr_run<2>(out, out_stride, nullptr, 0, batch, start, n, gates, params, state, workspace,
                       detach_walk != 0);
}

int pgx_adsr_gated_periodic_to(float *out, int64_t out_stride, int batch, int64_t start, int64_t n,
                               const pgx_gate_params *gates, const pgx_adsr_params *params, const double *state_in,
                               double *state_out, void *workspace, int detach_walk) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && gates && params && state_in && state_out && workspace,
                  "pgx_adsr_gated_periodic_to: null pointer");
    PGX_CHECK_ARG(n < (int64_t)1 << 30, "pgx_adsr_gated_periodic_to: block too long");
    PGX_CHECK_ARG(batch == 1 || out_stride >= n, "pgx_adsr_gated_periodic_to: stride too small");
    return adsr_run<2>(out, out_stride, nullptr, 0, batch, start, n, gates, params, const_cast<double *>(state_in),
                       workspace, detach_walk != 0, state_out);
}

int pgx_adsr_triggered(float *out, int64_t out_stride, const float *trig, int64_t trig_stride, int batch,
                       int64_t start, int64_t n, const pgx_adsr_params *params, double *state, void *workspace) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && trig && params && state && workspace, "pgx_adsr_triggered: null pointer");
    PGX_CHECK_ARG(n < (int64_t)1 << 30, "pgx_adsr_triggered: block too long");
    PGX_CHECK_ARG(batch == 1 || (out_stride >= n && trig_stride >= n), "pgx_adsr_triggered: stride too small");
    return adsr_launch<1>(out, out_stride, trig, trig_stride, batch, start, n, nullptr, params, state, workspace);
}

}  // extern "C"
