// pgx_lookup.hip -- gather / curve / sample-format kernels of the PEs either side of the hot path
// (SURVEY.md section 8f ranks 3-4): DelayPE's interpolated lookup, PiecewisePE, PCM16 conversion.
//
// All of them are one thread per output frame, HBM-bound, and reproduce the reference's numpy
// expression order (compiled with -ffp-contract=off) so that the linear/step paths are bit-exact.

#include "pgx_common.h"

namespace {

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------------
// DelayPE / interpolated_lookup (interpolated_lookup.py:28-77, delay_pe.py:170-216)
//   index = float64(start + i) - delay[i];  floor, fraction, clipped neighbours of the rendered
//   source window [win_start, win_start + win_len); out-of-extent indices -> 0.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_interp_lookup(float *out, const float *win, int64_t win_start, int64_t win_len, int channels, int64_t start,
                int64_t n, double delay_scalar, const float *delay, int cubic, int bounded, double ext_start,
                double ext_end) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const double index = (double)(start + i) - (delay ? (double)delay[i] : delay_scalar);
        const double fl = floor(index);
        const double t = index - fl;
        const int64_t p1 = (int64_t)fl - win_start;
        const bool oob = bounded && (index < ext_start || index >= ext_end);
        const int64_t last = win_len - 1;
        auto clip = [last](int64_t v) { return v < 0 ? 0 : (v > last ? last : v); };
        if (!cubic) {
            const float *a = win + clip(p1) * channels;
            const float *b = win + clip(p1 + 1) * channels;
            for (int c = 0; c < channels; ++c) {
                const double v = (1.0 - t) * (double)a[c] + t * (double)b[c];
                out[i * channels + c] = oob ? 0.0f : (float)v;
            }
        } else {
            const float *q0 = win + clip(p1 - 1) * channels;
            const float *q1 = win + clip(p1) * channels;
            const float *q2 = win + clip(p1 + 1) * channels;
            const float *q3 = win + clip(p1 + 2) * channels;
            const double t2 = t * t;
            const double t3 = t2 * t;
            for (int c = 0; c < channels; ++c) {
                const float p0 = q0[c], pa = q1[c], pb = q2[c], pc = q3[c];
                // float32 sub-expressions, exactly as numpy evaluates `2.0 * p1`, `-p0 + p2`, ... on float32 arrays
                const float k0 = 2.0f * pa;
                const float k1 = -p0 + pb;
                const float k2 = ((2.0f * p0 - 5.0f * pa) + 4.0f * pb) - pc;
                const float k3 = ((-p0 + 3.0f * pa) - 3.0f * pb) + pc;
                const double v = 0.5 * ((((double)k0 + (double)k1 * t) + (double)k2 * t2) + (double)k3 * t3);
                out[i * channels + c] = oob ? 0.0f : (float)v;
            }
        }
    }
}

// min / max of float64(start + i) - delay[i] over the block -> result[0..1]  (one workgroup)
__global__ void __launch_bounds__(kBlock)
k_index_range(double *result, const float *delay, int64_t start, int64_t n) {
    __shared__ double smin[kBlock], smax[kBlock];
    double lo = INFINITY, hi = -INFINITY;
    for (int64_t i = threadIdx.x; i < n; i += kBlock) {
        const double index = (double)(start + i) - (double)delay[i];
        lo = fmin(lo, index);            // np.min / np.max propagate NaN; a NaN delay is rejected on the host
        hi = fmax(hi, index);
        if (index != index) lo = hi = index;
    }
    smin[threadIdx.x] = lo;
    smax[threadIdx.x] = hi;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const double a = smin[threadIdx.x + s], b = smax[threadIdx.x + s];
            if (a != a || a < smin[threadIdx.x]) smin[threadIdx.x] = a;
            if (b != b || b > smax[threadIdx.x]) smax[threadIdx.x] = b;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        result[0] = smin[0];
        result[1] = smax[0];
    }
}

// min / max of a control stream, one (min, max) pair per workgroup (the host folds the few pairs): LadderPE sizes
// the warm-up of its time segments from the lowest cutoff and the highest resonance of the block.  NaN propagates.
__global__ void __launch_bounds__(kBlock)
k_stream_range(double *partials, const float *x, int64_t n) {
    __shared__ double smin[kBlock], smax[kBlock];
    double lo = INFINITY, hi = -INFINITY;
    bool nan = false;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double v = (double)x[i];
        lo = fmin(lo, v);
        hi = fmax(hi, v);
        nan = nan || v != v;
    }
    if (nan) lo = hi = NAN;
    smin[threadIdx.x] = lo;
    smax[threadIdx.x] = hi;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const double a = smin[threadIdx.x + s], b = smax[threadIdx.x + s];
            if (a != a || a < smin[threadIdx.x]) smin[threadIdx.x] = a;
            if (b != b || b > smax[threadIdx.x]) smax[threadIdx.x] = b;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 2 + 0] = smin[0];
        partials[blockIdx.x * 2 + 1] = smax[0];
    }
}

// ------------------------------------------------------------------------------------------------
// PiecewisePE (piecewise_pe.py:44-75, 164-229).  transition: 0 step, 1 linear, 2 exponential,
// 3 sigmoid, 4 constant_power.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_piecewise(float *out, int64_t start, int64_t n, int channels, const int64_t *times, const double *values,
            int count, int transition, int hold_first, int hold_last) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t t0 = times[0], t_last = times[count - 1];
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const int64_t s = start + i;
        float v = 0.0f;
        if (s < t0) {
            if (hold_first) v = (float)values[0];
        } else if (count == 1) {
            if (s == t0 || hold_last) v = (float)values[0];
        } else if (s >= t_last) {
            if (hold_last) v = (float)values[count - 1];
        } else {
            int lo = 0, hi = count;                        // upper_bound: first time > s
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (times[mid] <= s) lo = mid + 1;
                else hi = mid;
            }
            const int j = lo - 1;                          // times[j] <= s < times[j+1]
            const int64_t s0 = times[j], s1 = times[j + 1];
            const double v0 = values[j], v1 = values[j + 1];
            const double t = ((double)s - (double)s0) / (double)(s1 - s0);
            double r;
            if (transition == 0) {
                r = v0;
            } else if (transition == 2 && !(v0 <= 0.0 || v1 <= 0.0)) {
                r = v0 * pow(v1 / v0, t);
            } else if (transition == 3) {
                double x = 6.0 * (2.0 * t - 1.0);
                x = x < -20.0 ? -20.0 : (x > 20.0 ? 20.0 : x);
                r = v0 + (v1 - v0) * (1.0 / (1.0 + exp(-x)));
            } else if (transition == 4) {
                const double a = 0.5 * 3.141592653589793 * t;
                double sn, cs;
                pgx::pgx_sincos(a, sn, cs);
                r = v0 + (v1 - v0) * (v1 >= v0 ? sn : 1.0 - cs);
            } else {
                r = v0 + (v1 - v0) * t;
            }
            v = (float)r;
        }
        for (int c = 0; c < channels; ++c) out[i * channels + c] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// WAV sample formats: what the reference's writer does to a float32 sample on its way into a PCM_16 file.
// wav_writer_pe.py:67 opens the file through python-soundfile, whose SoundFile.__init__ switches libsndfile to
// clipping conversions (SFC_SET_CLIPPING = SF_TRUE); pcm_write_f2les then uses f2s_clip_array (src/pcm.c):
// scaled = x * 0x8000 in float arithmetic; scaled >= 0x7FFF -> 0x7FFF; scaled <= -0x8000 -> -0x8000; otherwise
// lrintf(scaled) in the default rounding mode (half to even).  Reading (s2f_array): x = s / 0x8000.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_f32_to_pcm16(int16_t *out, const float *in, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float scaled = in[i] * 32768.0f;
        int v;
        if (scaled >= 32767.0f) v = 32767;
        else if (scaled <= -32768.0f) v = -32768;
        else if (scaled != scaled) v = 0;                  // NaN: lrintf's result is unspecified; write silence
        else v = (int)rintf(scaled);                       // round half to even
        out[i] = (int16_t)v;
    }
}

__global__ void __launch_bounds__(kBlock)
k_pcm16_to_f32(float *out, const int16_t *in, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        out[i] = (float)in[i] * (1.0f / 32768.0f);
}

// ------------------------------------------------------------------------------------------------
// SpatialPE (spatial_pe.py:94-144, 179-214, 250-286): channel adaptation and stereo panning, float32
// arithmetic like the reference's numpy expressions (np.mean of a float32 row: sequential float32 sum / n).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float row_mean(const float *row, int from, int to) {
    float acc = row[from];
    for (int c = from + 1; c < to; ++c) acc = acc + row[c];
    return acc / (float)(to - from);
}

__global__ void __launch_bounds__(kBlock)
k_channel_adapt(float *out, const float *in, int64_t n, int src_ch, int out_ch) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float *x = in + i * src_ch;
        float *y = out + i * out_ch;
        if (src_ch == 1) {
            for (int c = 0; c < out_ch; ++c) y[c] = x[0];
        } else if (out_ch == 1) {
            y[0] = row_mean(x, 0, src_ch);
        } else if (src_ch == 2 && out_ch == 4) {
            const float cs = row_mean(x, 0, 2);
            y[0] = x[0]; y[1] = x[1]; y[2] = cs; y[3] = cs;
        } else if (src_ch == 4 && out_ch == 2) {
            y[0] = x[0]; y[1] = x[1];
        } else if (out_ch > src_ch) {
            for (int c = 0; c < src_ch; ++c) y[c] = x[c];
            for (int c = src_ch; c < out_ch; ++c) y[c] = x[src_ch - 1];
        } else {
            for (int c = 0; c < out_ch; ++c) y[c] = x[c];
            y[out_ch - 1] = y[out_ch - 1] + row_mean(x, out_ch, src_ch);
        }
    }
}

// mode 0: linear (L = 1 - pan, R = pan, pan = (az + 90) / 180); 1: constant power (cos / sin of (az + 90) / 2 deg)
__global__ void __launch_bounds__(kBlock)
k_pan(float *out, const float *in, int64_t n, int src_ch, float az_scalar, const float *az_stream, int mode) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float mono = row_mean(in + i * src_ch, 0, src_ch);
        float az = az_stream ? az_stream[i] : az_scalar;
        az = az < -90.0f ? -90.0f : (az > 90.0f ? 90.0f : az);
        float lg, rg;
        if (mode == 0) {
            const float pan = (az + 90.0f) / 180.0f;
            lg = 1.0f - pan;
            rg = pan;
        } else {
            const float ang = ((az + 90.0f) / 2.0f) * (float)(3.141592653589793 / 180.0);   // np.deg2rad in float32
            lg = cosf(ang);
            rg = sinf(ang);
        }
        out[i * 2 + 0] = mono * lg;
        out[i * 2 + 1] = mono * rg;
    }
}

__global__ void __launch_bounds__(kBlock)
k_mono_mean(float *out, const float *in, int64_t n, int src_ch) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        out[i] = row_mean(in + i * src_ch, 0, src_ch);
}

}  // namespace

extern "C" {

int pgx_channel_adapt(float *out, const float *in, int64_t n, int src_channels, int out_channels) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && src_channels >= 1 && out_channels >= 1, "pgx_channel_adapt: bad argument");
    hipLaunchKernelGGL(k_channel_adapt, dim3(pgx::grid_for(n, kBlock)), dim3(kBlock), 0, pgx::stream(), out, in, n,
                       src_channels, out_channels);
    PGX_LAUNCH_CHECK("k_channel_adapt");
    return PGX_OK;
}

int pgx_pan(float *out, const float *in, int64_t n, int src_channels, float azimuth, const float *azimuth_stream,
            int constant_power) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && src_channels >= 1, "pgx_pan: bad argument");
    hipLaunchKernelGGL(k_pan, dim3(pgx::grid_for(n, kBlock)), dim3(kBlock), 0, pgx::stream(), out, in, n,
                       src_channels, azimuth, azimuth_stream, constant_power ? 1 : 0);
    PGX_LAUNCH_CHECK("k_pan");
    return PGX_OK;
}

int pgx_mono_mean(float *out, const float *in, int64_t n, int src_channels) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && src_channels >= 1, "pgx_mono_mean: bad argument");
    hipLaunchKernelGGL(k_mono_mean, dim3(pgx::grid_for(n, kBlock)), dim3(kBlock), 0, pgx::stream(), out, in, n,
                       src_channels);
    PGX_LAUNCH_CHECK("k_mono_mean");
    return PGX_OK;
}


int pgx_interp_lookup(float *out, const float *window, int64_t window_start, int64_t window_len, int channels,
                      int64_t start, int64_t n, double delay_scalar, const float *delay, int cubic, int bounded,
                      double extent_start, double extent_end) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && window && window_len >= 1 && channels >= 1, "pgx_interp_lookup: bad argument");
    hipLaunchKernelGGL(k_interp_lookup, dim3(pgx::grid_for(n, kBlock)), dim3(kBlock), 0, pgx::stream(), out, window,
                       window_start, window_len, channels, start, n, delay_scalar, delay, cubic, bounded,
                       extent_start, extent_end);
    PGX_LAUNCH_CHECK("k_interp_lookup");
    return PGX_OK;
}

int pgx_index_range(double *result_dev, const float *delay, int64_t start, int64_t n) {
    PGX_REQUIRE_INIT();
    PGX_CHECK_ARG(result_dev && delay && n >= 1, "pgx_index_range: bad argument");
    hipLaunchKernelGGL(k_index_range, dim3(1), dim3(kBlock), 0, pgx::stream(), result_dev, delay, start, n);
    PGX_LAUNCH_CHECK("k_index_range");
    return PGX_OK;
}

int pgx_stream_range(double *partials_dev, int parts, const float *x, int64_t n) {
    PGX_REQUIRE_INIT();
    PGX_CHECK_ARG(partials_dev && x && n >= 1 && parts >= 1 && parts <= 1024, "pgx_stream_range: bad argument");
    hipLaunchKernelGGL(k_stream_range, dim3(parts), dim3(kBlock), 0, pgx::stream(), partials_dev, x, n);
    PGX_LAUNCH_CHECK("k_stream_range");
    return PGX_OK;
}

int pgx_piecewise(float *out, int64_t start, int64_t n, int channels, const int64_t *times, const double *values,
                  int count, int transition, int hold_first, int hold_last) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && times && values && count >= 1 && channels >= 1 && transition >= 0 && transition <= 4,
                  "pgx_piecewise: bad argument");
    hipLaunchKernelGGL(k_piecewise, dim3(pgx::grid_for(n, kBlock)), dim3(kBlock), 0, pgx::stream(), out, start, n,
                       channels, times, values, count, transition, hold_first, hold_last);
    PGX_LAUNCH_CHECK("k_piecewise");
    return PGX_OK;
}

int pgx_f32_to_pcm16(int16_t *out, const float *in, int64_t n_elems) {
    PGX_REQUIRE_INIT();
    if (n_elems <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in, "pgx_f32_to_pcm16: null pointer");
    hipLaunchKernelGGL(k_f32_to_pcm16, dim3(pgx::grid_for(n_elems, kBlock)), dim3(kBlock), 0, pgx::stream(), out, in,
                       n_elems);
    PGX_LAUNCH_CHECK("k_f32_to_pcm16");
    return PGX_OK;
}

int pgx_pcm16_to_f32(float *out, const int16_t *in, int64_t n_elems) {
    PGX_REQUIRE_INIT();
    if (n_elems <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in, "pgx_pcm16_to_f32: null pointer");
    hipLaunchKernelGGL(k_pcm16_to_f32, dim3(pgx::grid_for(n_elems, kBlock)), dim3(kBlock), 0, pgx::stream(), out, in,
                       n_elems);
    PGX_LAUNCH_CHECK("k_pcm16_to_f32");
    return PGX_OK;
}

}  // extern "C"
