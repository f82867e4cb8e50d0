// pgx_common.h -- internal helpers shared by the HIP translation units of libpygmu_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <string>

#include "pygmu_hip.h"

namespace pgx {

// Thread-local last-error message (pgx_last_error()).
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

// Library stream (created by pgx_init).  Returns nullptr before init.
hipStream_t stream();
hipStream_t main_stream();   // the stream `stream()` returns outside a fork
bool initialised();
int device_index();

}  // namespace pgx

#define PGX_REQUIRE_INIT()                                                        \
    do {                                                                          \
        if (!pgx::initialised())                                                  \
            return pgx::fail(PGX_ERR_NOT_INIT, "pgx_init() has not been called"); \
    } while (0)

// Branch weights for the instruction counts of tools/isa_count.py.  That tool compiles the kernels with
// -DPGX_COUNT_STEADY and counts the float64 instructions of their loop bodies for the FP64-VALU roofline: under that
// switch a branch marked PGX_COLD (a workgroup's first tile, the guarded re-run after a singularity, the one thread that
// captures the carried state, unaligned edges) is compiled out and one marked PGX_HOT is taken unconditionally, so the
// bodies hold the steady-state path only.  The library itself is never built with the switch: there both are `(cond)`.
#ifdef PGX_COUNT_STEADY
#define PGX_COLD(cond) (false)
#define PGX_HOT(cond) (true)
#else
#define PGX_COLD(cond) (cond)
#define PGX_HOT(cond) (cond)
#endif

#define PGX_CHECK_ARG(cond, msg)                                     \
    do {                                                             \
        if (!(cond)) return pgx::fail(PGX_ERR_INVALID, (msg));       \
    } while (0)

#define PGX_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t _e = (call);                                                         \
        if (_e != hipSuccess) {                                                         \
            return pgx::fail(_e == hipErrorOutOfMemory ? PGX_ERR_NOMEM : PGX_ERR_RUNTIME, \
                             std::string(#call) + ": " + hipGetErrorString(_e));        \
        }                                                                               \
    } while (0)

// Launch-error check (does not synchronise).
#define PGX_LAUNCH_CHECK(name)                                                        \
    do {                                                                              \
        hipError_t _e = hipGetLastError();                                            \
        if (_e != hipSuccess)                                                         \
            return pgx::fail(PGX_ERR_RUNTIME, std::string(name) + " launch: " +       \
                                                  hipGetErrorString(_e));             \
    } while (0)

namespace pgx {

constexpr int kWave = 64;      // gfx950 wavefront
constexpr int kNumCU = 256;    // MI355X

__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------
// float64 sine / cosine for the oscillator kernels.
// Cody-Waite reduction by pi with a 33+33+53-bit split, each step one fused multiply-add: x - q*PI_A is exact
// (a multiple of 2^-31 below 4 in magnitude) for |q| < 2^31, the two further steps each round once at the size
// of the reduced argument.  Then the Taylor polynomials on |r| <= pi/2 (degree 21 / 22: truncation < 2e-18).
// Error ~1-2 ulp, like the libm / SVML routines behind the reference's np.sin (checked on the CPU against sinl
// on 5e7 random arguments per decade up to 4e9 rad: 2.2e-16 absolute, no float32 result differing from libm's);
// arguments beyond kSinFastRange = 2e9 rad (a 440 Hz tone after 200 hours) fall back to the ocml routine.
constexpr double kSinFastRange = 2.0e9;
#ifdef __HIPCC__
// a / b, correctly rounded, for a divisor whose reciprocal y = RN(1/b) the caller made once (Markstein: a
// faithful quotient corrected by its exact fused residual times a correctly rounded reciprocal rounds like the
// division; the first correction makes q faithful, the second rounds it).  Five multiply-adds instead of the
// IEEE sequence around v_rcp_f64; 5.9e9 numerators against 16 sample rates on the CPU: no difference from `/`.
__device__ __forceinline__ double pgx_div_by(double a, double b, double y) {
    double q = a * y;
    double r = __builtin_fma(-b, q, a);
    q = __builtin_fma(r, y, q);
    r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}

__device__ __forceinline__ double pgx_reduce_pi(double x, int &qi) {
    const double PI_A = 0x1.921fb54400000p+1, PI_B = 0x1.0b4611a600000p-33, PI_C = 0x1.3198a2e037073p-68;
    // q = rint(x / pi) by the add-and-subtract of 1.5 * 2^52 (the sum has no fraction bits left: it rounds to
    // the nearest-even integer exactly as rint does, for |x / pi| < 2^51), its parity from the sum's low word --
    // two full-rate adds where v_rndne_f64 and v_cvt_i32_f64 run at a quarter of the rate
    const double MAGIC = 0x1.8p+52;
    const double qm = x * 0x1.45f306dc9c883p-2 + MAGIC;
    const double q = qm - MAGIC;
    double r = __builtin_fma(-q, PI_A, x);
    r = __builtin_fma(-q, PI_B, r);
    r = __builtin_fma(-q, PI_C, r);
    qi = __double2loint(qm);                                     // q mod 2^32 (|q| < 2^31): only its parity is used
    return r;
}
__device__ __forceinline__ double pgx_sin_poly(double r) {       // sin(r), |r| <= pi/2
    const double s = r * r;
    double p = -1.0 / 51090942171709440000.0;                    // -1/21!
    p = __builtin_fma(p, s, 1.0 / 121645100408832000.0);         // +1/19!
    p = __builtin_fma(p, s, -1.0 / 355687428096000.0);           // -1/17!
    p = __builtin_fma(p, s, 1.0 / 1307674368000.0);              // +1/15!
    p = __builtin_fma(p, s, -1.0 / 6227020800.0);                // -1/13!
    p = __builtin_fma(p, s, 1.0 / 39916800.0);                   // +1/11!
    p = __builtin_fma(p, s, -1.0 / 362880.0);                    // -1/9!
    p = __builtin_fma(p, s, 1.0 / 5040.0);                       // +1/7!
    p = __builtin_fma(p, s, -1.0 / 120.0);                       // -1/5!
    p = __builtin_fma(p, s, 1.0 / 6.0);                          // +1/3!
    // p = 1/3! - s/5! + s^2/7! - ... ; sin(r) = r - r^3 * p
    return __builtin_fma(-(r * s), p, r);
}
__device__ __forceinline__ double pgx_cos_poly(double r) {       // cos(r), |r| <= pi/2
    const double s = r * r;
    double p = 1.0 / 1124000727777607680000.0;                   // +1/22!
    p = __builtin_fma(p, s, -1.0 / 2432902008176640000.0);       // -1/20!
    p = __builtin_fma(p, s, 1.0 / 6402373705728000.0);           // +1/18!
    p = __builtin_fma(p, s, -1.0 / 20922789888000.0);            // -1/16!
    p = __builtin_fma(p, s, 1.0 / 87178291200.0);                // +1/14!
    p = __builtin_fma(p, s, -1.0 / 479001600.0);                 // -1/12!
    p = __builtin_fma(p, s, 1.0 / 3628800.0);                    // +1/10!
    p = __builtin_fma(p, s, -1.0 / 40320.0);                     // -1/8!
    p = __builtin_fma(p, s, 1.0 / 720.0);                        // +1/6!
    p = __builtin_fma(p, s, -1.0 / 24.0);                        // -1/4!
    p = __builtin_fma(p, s, 0.5);                                // +1/2!
    return __builtin_fma(-s, p, 1.0);                            // 1 - s*p
}
// np.mod(a, 1.0) (result in [0, 1], sign of the divisor): a - floor(a) is the same value --
// fmod(a,1) is exact, and numpy's "+1 if negative" rounds once, exactly like this subtraction.
__device__ __forceinline__ double pgx_mod1(double a) {
    const double r = a - floor(a);
    return (r == 0.0) ? 0.0 : r;      // +0 for integers and for -0.0
}
// a / b with a Newton-refined hardware reciprocal (<= 1 ulp; ~10 instructions instead of the ~30 of the
// correctly rounded IEEE sequence).  For finite b away from the subnormal range.
__device__ __forceinline__ double pgx_div_fast(double a, double b) {
    double y = __builtin_amdgcn_rcp(b);
    y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
    const double q = a * y;
    return __builtin_fma(__builtin_fma(-q, b, a), y, q);
}

// The same with ONE Newton step on the reciprocal: v_rcp_f64 is good to 2^-24.4 (tools/microbench/rcp_acc.hip), one
// step to 2^-48.7, and the residual correction then leaves ~2^-97 + the final rounding: within an ulp of the
// quotient (the second step only decides ties of the last bit).  Two instructions less on a per-sample path.
__device__ __forceinline__ double pgx_div_fast1(double a, double b) {
    double y = __builtin_amdgcn_rcp(b);
    y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
    const double q = a * y;
    return __builtin_fma(__builtin_fma(-q, b, a), y, q);
}

// tanh for the ladder's feedback loop: branch-free, ~45 instructions instead of libm's ~110 on the
// critical path of a strictly sequential recurrence.  e = exp(-2|x|) by Cody-Waite reduction
// (ln2 = hi + lo) and a degree-13 Taylor polynomial on |r| <= ln2/2, tanh = (1 - e) / (1 + e) with a
// Newton-refined reciprocal.  Error: <= 2 ulp for |x| >= 0.25; below that the cancellation in 1 - e
// leaves an ABSOLUTE error <= 1.2e-16 (relative 1.2e-16/|x|), which is what a filter state needs.
__device__ __forceinline__ double pgx_tanh(double x) {
    double ax = fabs(x);
    ax = ax < 40.0 ? ax : 40.0;                                  // e < 2^-115: tanh == 1; NaN -> 40 below
    const double t = ax + ax;
    const double kf = rint(t * 1.4426950408889634);              // t / ln2
    double r = __builtin_fma(-kf, 6.93147180369123816490e-01, t);
    r = __builtin_fma(-kf, 1.90821492927058770002e-10, r);       // r in [-ln2/2, ln2/2]
    // exp(-r)
    double p = -1.0 / 6227020800.0;                              // -1/13!
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, -1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, -1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, -1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, -1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, -1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, -1.0);
    p = __builtin_fma(p, r, 1.0);
    const double e = ldexp(p, -(int)kf);                         // exp(-2|x|) in (0, 1]
    const double num = 1.0 - e, den = 1.0 + e;
    double y = __builtin_amdgcn_rcp(den);
    y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
    double q = num * y;
    q = __builtin_fma(__builtin_fma(-q, den, num), y, q);
    q = (x != x) ? x : q;                                        // NaN in, NaN out
    return copysign(q, x);
}

__device__ __forceinline__ double pgx_sin(double x) {
    if (!(fabs(x) < kSinFastRange)) return sin(x);
    int qi;
    const double r = pgx_reduce_pi(x, qi);
    const double v = pgx_sin_poly(r);
    return (qi & 1) ? -v : v;
}
// The same routine for callers that guarantee |x| < kSinFastRange: without the fallback branch the code is one basic
// block, so the compiler interleaves several evaluations (the oscillators evaluate 16 per thread and tile; with
// the branch each one ran as a lone dependent chain).  Same bits as pgx_sin on that range.
__device__ __forceinline__ double pgx_sin_bounded(double x) {
    int qi;
    const double r = pgx_reduce_pi(x, qi);
    const double v = pgx_sin_poly(r);
    return (qi & 1) ? -v : v;
}
__device__ __forceinline__ void pgx_sincos_bounded(double x, double &sn, double &cs) {   // |x| < kSinFastRange (or NaN)
    int qi;
    const double r = pgx_reduce_pi(x, qi);
    const double a = pgx_sin_poly(r), b = pgx_cos_poly(r);
    sn = (qi & 1) ? -a : a;
    cs = (qi & 1) ? -b : b;
}
__device__ __forceinline__ void pgx_sincos(double x, double &sn, double &cs) {
    if (!(fabs(x) < kSinFastRange)) {
        sn = sin(x);
        cs = cos(x);
        return;
    }
    int qi;
    const double r = pgx_reduce_pi(x, qi);
    const double a = pgx_sin_poly(r), b = pgx_cos_poly(r);
    sn = (qi & 1) ? -a : a;
    cs = (qi & 1) ? -b : b;
}
#endif

// Grid size for a grid-stride elementwise kernel: a workgroup per `block` items, at most 128 per CU (PGX_GRID_CAP).
// (8 per CU until the end of round 4 -- enough to fill the chip, but a launch of a look-ahead window's size then walks its
// buffer in 2 048 strided runs, which the memory system takes slower than many short workgroups each writing ONE contiguous
// piece: fill 134 M floats 4.7 -> 5.9 TB/s, gain 5.0 -> 5.7; at 11 M frames -- a 256-block window of 44 100 frames -- no
// difference.  tools/microbench/fill_rate.hip: hipMemsetAsync 6.5 TB/s, one 16 KB chunk per workgroup 6.0, the strided loop 4.6)
inline int grid_for(int64_t work_items, int block) {
    static const int per_cu = getenv("PGX_GRID_CAP") ? atoi(getenv("PGX_GRID_CAP")) : 128;
    int64_t g = ceil_div(work_items, block);
    if (g < 1) g = 1;
    if (g > (int64_t)kNumCU * per_cu) g = (int64_t)kNumCU * per_cu;
    return (int)g;
}

}  // namespace pgx
