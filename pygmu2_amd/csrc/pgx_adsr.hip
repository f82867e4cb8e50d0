// pgx_adsr.hip -- AdsrGatedPE / AdsrTriggeredPE (adsr_pe.py:124-196, :279-335), bit-exact and
// wave-parallel.
//
// The reference is a per-sample Python state machine: emit the level, react to the gate edge /
// trigger, then `env += slope` with clamping.  The float64 accumulation must be reproduced
// exactly -- not just to ~1e-16: whether the attack ends after 480 or 481 steps depends on the
// rounding of the running sum -- so `env_k = env_0 + k*slope` is NOT an acceptable shortcut.
//
// What makes it parallel anyway: while the level stays inside one binade [2^e, 2^(e+1)) every
// addition of the constant slope d rounds to the same grid of spacing u = 2^(e-52).  Writing
// env = E*u and |d| = (D + r)*u with integer E, D and 0 <= r < 1 (all exact power-of-two
// scalings), round-to-nearest gives E' = E +/- (D + [r > 1/2]) as long as no tie (r == 1/2)
// occurs and the sum does not leave the binade.  Inside such a "run" the level is an exact
// integer progression env_t = env_0 + t*dq with dq = +/-(D + [r > 1/2])*u, every term of which is
// representable, so the 64 lanes of a wave emit 64 consecutive samples at once.  Binade
// crossings, ties, threshold crossings (>= 1, <= sustain, <= 0), zero levels and gate edges are
// handled by taking ONE literal reference step.  An ADSR cycle is a few dozen runs, so a voice
// costs about one short wave-iteration per 64 samples instead of 64 dependent iterations.
//
// One wave per envelope; gate/trigger rows are read and the output rows written fully coalesced.

#include "pgx_common.h"

namespace {

constexpr int kIdle = 0, kAttack = 1, kDecay = 2, kSustain = 3, kRelease = 4;
constexpr long long kInf = 1LL << 62;
constexpr long long kTwo52 = 1LL << 52;
constexpr long long kTwo53 = 1LL << 53;

struct AdsrRun {
    double dq;          // exact per-sample increment of the current run
    long long left;     // regular steps still available in this run (0 = must re-plan)
};

// Plan a run from (state, env): how many steps can be taken as env += dq exactly.
__device__ __forceinline__ AdsrRun adsr_plan(int s, double env, const pgx_adsr_params &p, bool triggered,
                                             long long now, long long ends_at) {
    AdsrRun run{0.0, 0};
    if (s == kIdle) {
        if (env == 0.0) run.left = kInf;
        return run;
    }
    if (s == kSustain) {
        if (env == p.sustain_level) {
            if (!triggered) run.left = kInf;
            else if (now < ends_at) run.left = ends_at - now;
        }
        return run;
    }
    const double d = (s == kAttack) ? p.attack_dvdt : (s == kDecay ? p.decay_dvdt : p.release_dvdt);
    // runs are planned only for the ordinary slope signs (attack up, decay/release down or flat);
    // anything else (e.g. sustain_level > 1) is stepped literally
    if ((s == kAttack) ? (d < 0.0) : (d > 0.0)) return run;
    if (d != d) return run;
    if (!(env >= 1e-290) || !(env < 1e290)) return run;           // zero, negative, tiny, inf, nan
    const int e = ilogb(env);
    if (s == kAttack && e >= 0) return run;                        // env >= 1: the clamp fires next
    const double u = ldexp(1.0, e - 52);
    const double q = fabs(d) / u;                                  // exact (power-of-two scaling)
    if (!(q < 9007199254740992.0)) return run;                     // |d| >= 2^(e+1): leaves the binade
    const double Dd = floor(q);
    const double r = q - Dd;
    if (r == 0.5) return run;                                      // tie: round-half-even, step literally
    const long long D = (long long)Dd;
    const long long Dq = D + (r > 0.5 ? 1 : 0);
    const long long E = (long long)(env / u);                      // in [2^52, 2^53), exact
    if (Dq == 0) {                                                 // |d| < u/2: the level cannot move
        if (s == kDecay && env <= p.sustain_level) return run;
        run.left = kInf;
        return run;
    }
    if (d > 0.0) {
        // step t is regular while E_{t-1} + D + 1 <= 2^53 - 1 (stays strictly inside the binade)
        const long long num = kTwo53 - 2 - D - E;
        if (num < 0) return run;
        run.left = num / Dq + 1;
        run.dq = (double)Dq * u;
    } else {
        // step t is regular while E_{t-1} - D - 1 >= 2^52
        const long long num = E - D - 1 - kTwo52;
        if (num < 0) return run;
        long long k = num / Dq + 1;
        // and while the new level stays above the clamp threshold: E_t >= floor(thr/u) + 1
        const double thr = (s == kDecay) ? p.sustain_level : 0.0;
        const double tq = thr / u;
        if (!(tq < 9007199254740992.0)) return run;
        const long long F = (long long)floor(tq);
        const long long room = E - F - 1;
        if (room < 0) return run;
        const long long kt = room / Dq;
        if (kt < k) k = kt;
        if (k <= 0) return run;
        run.left = k;
        run.dq = -((double)Dq * u);
    }
    return run;
}

// One literal reference step (after the level has been emitted for this sample).
__device__ __forceinline__ void adsr_step(int &s, double &env, const pgx_adsr_params &p, bool triggered,
                                          long long now, long long &ends_at) {
    if (s == kIdle) {
        env = 0.0;
    } else if (s == kAttack) {
        env += p.attack_dvdt;
        if (env >= 1.0) { env = 1.0; s = kDecay; }
    } else if (s == kDecay) {
        env += p.decay_dvdt;
        if (env <= p.sustain_level) {
            env = p.sustain_level;
            if (triggered) ends_at = now + p.sustain_samples;
            s = kSustain;
        }
    } else if (s == kSustain) {
        env = p.sustain_level;
        if (triggered && now >= ends_at) s = kRelease;
    } else {
        env += p.release_dvdt;
        if (env <= 0.0) { env = 0.0; s = kIdle; }
    }
}

template <bool TRIG>
__global__ void __launch_bounds__(256)
k_adsr(float *out, int64_t out_stride, const float *ctl, int64_t ctl_stride, int batch, int64_t start,
       int64_t n, const pgx_adsr_params *params, double *state) {
    const int lane = threadIdx.x & 63;
    const int inst = blockIdx.x * 4 + (threadIdx.x >> 6);          // one wave per envelope
    if (inst >= batch) return;
    const pgx_adsr_params p = params[inst];
    const float *g = ctl + (int64_t)inst * ctl_stride;
    float *o = out + (int64_t)inst * out_stride;
    double *st = state + (int64_t)inst * 3;

    int s = (int)st[0];
    double env = st[1];
    float prev_gate = TRIG ? 0.0f : (float)st[2];
    long long ends_at = TRIG ? (long long)st[2] : 0;
    AdsrRun run{0.0, 0};

    for (int64_t i0 = 0; i0 < n; i0 += 64) {
        const int nvalid = (n - i0 < 64) ? (int)(n - i0) : 64;
        const bool valid = lane < nvalid;
        const float cur = valid ? g[i0 + lane] : 0.0f;
        unsigned long long amask, rmask;
        if (TRIG) {
            amask = __ballot(valid && cur > 0.0f);
            rmask = 0ull;
        } else {
            float pv = __shfl_up(cur, 1, 64);
            if (lane == 0) pv = prev_gate;
            amask = __ballot(valid && pv == 0.0f && cur == 1.0f);
            rmask = __ballot(valid && pv == 1.0f && cur == 0.0f);
            prev_gate = __shfl(cur, nvalid - 1, 64);
        }
        unsigned long long emask = amask | rmask;

        double mine = 0.0;
        int a = 0;
        while (a < nvalid) {
            if ((emask >> a) & 1ull) {                             // gate edge / trigger on this sample
                s = ((amask >> a) & 1ull) ? kAttack : kRelease;
                run.left = 0;
                emask &= ~(1ull << a);
            }
            const unsigned long long later = emask & ~((2ull << a) - 1ull);
            int limit = later ? (__ffsll((long long)later) - 1) : nvalid;
            if (limit > nvalid) limit = nvalid;
            const long long now = (long long)(start + i0 + a);
            if (run.left == 0) run = adsr_plan(s, env, p, TRIG, now, ends_at);
            if (run.left == 0) {                                   // literal step for one sample
                if (lane == a) mine = env;
                adsr_step(s, env, p, TRIG, now, ends_at);
                a += 1;
                continue;
            }
            long long take = limit - a;
            if (run.left < take) take = run.left;
            const int t = lane - a;
            if (t >= 0 && t < (int)take) mine = env + (double)t * run.dq;   // exact (see header)
            env = env + (double)take * run.dq;
            if (run.left < kInf) run.left -= take;
            a += (int)take;
        }
        if (valid) o[i0 + lane] = (float)mine;
    }
    if (lane == 0) {
        st[0] = (double)s;
        st[1] = env;
        st[2] = TRIG ? (double)ends_at : (double)prev_gate;
    }
}

}  // namespace

extern "C" {

int pgx_adsr_gated(float *out, int64_t out_stride, const float *gate, int64_t gate_stride, int batch, int64_t n,
                   const pgx_adsr_params *params, double *state) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && gate && params && state, "pgx_adsr_gated: null pointer");
    PGX_CHECK_ARG(batch == 1 || (out_stride >= n && gate_stride >= n), "pgx_adsr_gated: stride too small");
    hipLaunchKernelGGL(k_adsr<false>, dim3((batch + 3) / 4), dim3(256), 0, pgx::stream(), out, out_stride, gate,
                       gate_stride, batch, (int64_t)0, n, params, state);
    PGX_LAUNCH_CHECK("k_adsr<gated>");
    return PGX_OK;
}

int pgx_adsr_triggered(float *out, int64_t out_stride, const float *trig, int64_t trig_stride, int batch,
                       int64_t start, int64_t n, const pgx_adsr_params *params, double *state) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && trig && params && state, "pgx_adsr_triggered: null pointer");
    PGX_CHECK_ARG(batch == 1 || (out_stride >= n && trig_stride >= n), "pgx_adsr_triggered: stride too small");
    hipLaunchKernelGGL(k_adsr<true>, dim3((batch + 3) / 4), dim3(256), 0, pgx::stream(), out, out_stride, trig,
                       trig_stride, batch, start, n, params, state);
    PGX_LAUNCH_CHECK("k_adsr<triggered>");
    return PGX_OK;
}

}  // extern "C"
