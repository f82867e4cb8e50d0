// pgx_dynamics.hip -- the remaining PEs of the reference's throughput suite (benchmarks/benchmark_pes.py:309-350):
//   LoopPE      (loop_pe.py:159-232)      modular gather over the loop region + linear crossfade, bit-exact
//   WindowPE    (window_pe.py:118-254)    centred sliding max / min / mean / RMS
//   DynamicsPE  (dynamics_pe.py:190-372)  gain computer on an envelope (compress / limit / expand / gate)
// CompressorPE / LimiterPE / ExpanderPE are host compositions of CachePE + EnvelopePE + DynamicsPE
// (compressor_pe.py), so they add no kernels.
//
// All three are HBM-streaming: 4C B/frame read + 4C B/frame written (DynamicsPE reads the envelope too).

#include <hip/hip_runtime.h>

#include "pgx_common.h"

namespace {

constexpr int kBlock = 256;

// ================================================================================================ LoopPE
// out[i] = loop[(start + i) mod L]; the last `xf` positions of the loop fade into its first `xf` ones with the
// reference's float64 weights  fade_in = fade_pos / xf,  fade_out = 1 - fade_pos / xf  (loop_pe.py:203-229).
__global__ void __launch_bounds__(kBlock)
k_loop(float *out, const float *loop, int64_t start, int64_t n, int channels, int64_t loop_len, int64_t total,
       int64_t xf) {
    const int64_t elems = n * channels;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < elems; e += stride) {
        const int64_t i = e / channels;
        const int c = (int)(e - i * channels);
        const int64_t abs_i = start + i;
        if (total >= 0 && abs_i >= total) {                   // past count * L: silence (loop_pe.py:176-181)
            out[e] = 0.0f;
            continue;
        }
        int64_t pos = abs_i % loop_len;
        if (pos < 0) pos += loop_len;                         // numpy's modulo takes the divisor's sign
        float v = loop[pos * channels + c];
        const int64_t thr = loop_len - xf;
        if (xf > 0 && pos >= thr) {
            const int64_t fp = pos - thr;
            const double q = (double)fp / (double)xf;
            const double fade_out = 1.0 - q;
            const double a = (double)v * fade_out;
            const double b = (double)loop[fp * channels + c] * q;
            v = (float)(a + b);
        }
        out[e] = v;
    }
}

// ================================================================================================ WindowPE
// Window of w = 2h + 1 samples centred on each output; the caller hands over the source rendered from
// start - h with n + 2h frames (window_pe.py:139-142), so output i covers padded frames [i, i + w).
// Two levels: 64-frame block statistics, then per output the partial head, the whole blocks and the partial tail
// (<= 126 + w/64 terms instead of w).  MODE 0 max, 1 min, 2 mean, 3 rms.
constexpr int kWinBlock = 64;

template <int MODE>
__device__ __forceinline__ double win_load(const float *x, int64_t f, int channels, int c, bool rectify) {
    double v = (double)x[f * channels + c];
    if (rectify) v = fabs(v);
    return MODE == 3 ? v * v : v;
}
template <int MODE>
__device__ __forceinline__ double win_fold(double acc, double v) {
    if (MODE == 0) return v > acc ? v : acc;
    if (MODE == 1) return v < acc ? v : acc;
    return acc + v;
}
template <int MODE>
__device__ __forceinline__ double win_identity() {
    return MODE == 0 ? -__builtin_inf() : (MODE == 1 ? __builtin_inf() : 0.0);
}

template <int MODE>
__global__ void __launch_bounds__(kBlock)
k_window_blocks(double *blocks, const float *x, int64_t padded, int channels, int rectify) {
    const int lane = threadIdx.x & 63;
    const int64_t nblocks = padded / kWinBlock;               // whole blocks only
    const int64_t item = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);   // one wave per (block, channel)
    if (item >= nblocks * channels) return;
    const int64_t b = item / channels;
    const int c = (int)(item - b * channels);
    double acc = win_load<MODE>(x, b * kWinBlock + lane, channels, c, rectify);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc = win_fold<MODE>(acc, __shfl_down(acc, d, 64));
    if (lane == 0) blocks[item] = acc;
}

template <int MODE>
__global__ void __launch_bounds__(kBlock)
k_window_apply(float *out, const float *x, const double *blocks, int64_t n, int channels, int64_t w, int rectify) {
    const int64_t total = n * channels;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t i = e / channels;
        const int c = (int)(e - i * channels);
        const int64_t lo = i, hi = i + w;                     // [lo, hi) in padded frames
        const int64_t b0 = (lo + kWinBlock - 1) / kWinBlock, b1 = hi / kWinBlock;       // whole blocks [b0, b1)
        double acc = win_identity<MODE>();
        if (b0 < b1) {
            for (int64_t f = lo; f < b0 * kWinBlock; ++f) acc = win_fold<MODE>(acc, win_load<MODE>(x, f, channels, c, rectify));
            for (int64_t b = b0; b < b1; ++b) acc = win_fold<MODE>(acc, blocks[b * channels + c]);
            for (int64_t f = b1 * kWinBlock; f < hi; ++f) acc = win_fold<MODE>(acc, win_load<MODE>(x, f, channels, c, rectify));
        } else {
            for (int64_t f = lo; f < hi; ++f) acc = win_fold<MODE>(acc, win_load<MODE>(x, f, channels, c, rectify));
        }
        if (MODE == 2) acc = acc / (double)w;
        if (MODE == 3) acc = sqrt(acc / (double)w);
        out[e] = (float)acc;
    }
}

// ================================================================================================ DynamicsPE
// numpy types the whole gain computer float32 (the envelope is float32 and Python scalars are weak), except the
// hard-knee gate whose np.where(cond, range, 0.0) is float64: both are followed here, with log10 / 10**x taken
// in float64 and rounded once (numpy's float32 routines may differ from that by an ulp or two).
__device__ __forceinline__ float dyn_gain_linear(float env, const pgx_dynamics_params &p) {
    const float level = 20.0f * (float)log10((double)fmaxf(env, 1e-10f));
    if (p.mode == 3 && !p.soft) {                             // gate, hard knee: float64 from here on
        const double g = ((level < p.threshold) ? p.gate_range_d : 0.0) + p.makeup_d;
        return (float)pow(10.0, g / 20.0);
    }
    float gain = 0.0f;
    if (p.mode == 0 || p.mode == 1) {                         // compress / limit (ratio = inf)
        const float over = level - p.threshold;
        const float full = p.mode == 1 ? -over : over * p.slope;
        if (!p.soft) {
            gain = (level > p.threshold) ? full : 0.0f;
        } else {
            const float x = over + p.half_knee;
            const float sq = x * x;
            const float kg = p.mode == 1 ? (-sq) / p.two_knee : (p.slope * sq) / p.two_knee;
            gain = (level < p.knee_lo) ? 0.0f : ((level > p.knee_hi) ? full : kg);
        }
    } else if (p.mode == 2) {                                 // expand
        const float under = p.threshold - level;
        const float full = (-under) * p.slope;                // slope = ratio - 1
        if (!p.soft) {
            gain = (level < p.threshold) ? full : 0.0f;
        } else {
            const float x = p.knee_hi - level;
            const float kg = (p.neg_slope * (x * x)) / p.two_knee;
            gain = (level > p.knee_hi) ? 0.0f : ((level < p.knee_lo) ? full : kg);
        }
    } else {                                                  // gate, soft knee
        const float t = (p.knee_hi - level) / p.knee;
        gain = (level > p.knee_hi) ? 0.0f : ((level < p.knee_lo) ? p.gate_range : t * p.gate_range);
    }
    if (p.wide_makeup) return (float)pow(10.0, ((double)gain + p.makeup_d) / 20.0);
    gain = gain + p.makeup;
    return (float)pow(10.0, (double)(gain / 20.0f));
}

__global__ void __launch_bounds__(kBlock)
k_dynamics(float *out, const float *audio, const float *env, int64_t n, int channels, int env_channels,
           const pgx_dynamics_params *params) {
    const pgx_dynamics_params p = params[0];
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    // envelope layout (dynamics_pe.py:346-356): linked -> max over its channels; mono or mismatched -> channel 0
    const bool linked = p.stereo_link && env_channels > 1;
    const bool shared = linked || env_channels != channels;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float g = 0.0f;
        if (shared) {
            float e = env[i * env_channels];
            if (linked)
                for (int c = 1; c < env_channels; ++c) e = fmaxf(e, env[i * env_channels + c]);
            g = dyn_gain_linear(e, p);
        }
        for (int c = 0; c < channels; ++c) {
            if (!shared) g = dyn_gain_linear(env[i * env_channels + c], p);
            out[i * channels + c] = audio[i * channels + c] * g;
        }
    }
}

}  // namespace

extern "C" {

int pgx_loop(float *out, const float *loop, int64_t start, int64_t n, int channels, int64_t loop_len,
             int64_t total_len, int64_t crossfade) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && loop && channels >= 1 && loop_len >= 1 && crossfade >= 0 && crossfade <= loop_len / 2,
                  "pgx_loop: bad argument");
    hipLaunchKernelGGL(k_loop, dim3(pgx::grid_for(n * channels, kBlock)), dim3(kBlock), 0, pgx::stream(), out, loop,
                       start, n, channels, loop_len, total_len, crossfade);
    PGX_LAUNCH_CHECK("k_loop");
    return PGX_OK;
}

size_t pgx_window_workspace_bytes(int64_t n, int channels, int64_t half_window) {
    if (n <= 0 || channels <= 0 || half_window < 0) return 0;
    return (size_t)((n + 2 * half_window) / kWinBlock + 1) * channels * sizeof(double);
}

int pgx_window(float *out, const float *padded, int64_t n, int channels, int64_t half_window, int mode, int rectify,
               void *workspace) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && padded && workspace && channels >= 1 && half_window >= 0 && mode >= 0 && mode <= 3,
                  "pgx_window: bad argument");
    const int64_t w = 2 * half_window + 1, plen = n + 2 * half_window;
    double *blocks = (double *)workspace;
    const int64_t nblk = plen / kWinBlock;
#define PGX_WINDOW(MODE)                                                                                         \
    do {                                                                                                         \
        if (nblk > 0)                                                                                            \
            hipLaunchKernelGGL(k_window_blocks<MODE>, dim3((unsigned)((nblk * channels + 3) / 4)), dim3(kBlock), 0, \
                               pgx::stream(), blocks, padded, plen, channels, rectify);                          \
        hipLaunchKernelGGL(k_window_apply<MODE>, dim3(pgx::grid_for(n * channels, kBlock)), dim3(kBlock), 0,     \
                           pgx::stream(), out, padded, blocks, n, channels, w, rectify);                         \
    } while (0)
    switch (mode) {
    case 0: PGX_WINDOW(0); break;
    case 1: PGX_WINDOW(1); break;
    case 2: PGX_WINDOW(2); break;
    default: PGX_WINDOW(3); break;
    }
#undef PGX_WINDOW
    PGX_LAUNCH_CHECK("k_window");
    return PGX_OK;
}

int pgx_dynamics(float *out, const float *audio, const float *envelope, int64_t n, int channels, int env_channels,
                 const pgx_dynamics_params *params) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && audio && envelope && params && channels >= 1 && env_channels >= 1,
                  "pgx_dynamics: bad argument");
    hipLaunchKernelGGL(k_dynamics, dim3(pgx::grid_for(n, kBlock)), dim3(kBlock), 0, pgx::stream(), out, audio,
                       envelope, n, channels, env_channels, params);
    PGX_LAUNCH_CHECK("k_dynamics");
    return PGX_OK;
}

}  // extern "C"
