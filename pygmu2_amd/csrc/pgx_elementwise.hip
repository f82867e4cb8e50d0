// pgx_elementwise.hip -- stateless per-sample kernels: sources, SinePE (pure), GainPE, MixPE,
// gates/triggers, SuperSaw voice sum.
//
// All of these are HBM-streaming kernels: each lane owns 4 consecutive float32 output
// elements (16 B/lane, 1 KiB per wave store), grids are capped at 256 CUs x 8 blocks and
// grid-stride over the rest.  Compiled with -ffp-contract=off so float64 expressions are
// rounded operation by operation like the reference's numpy code.

#include "pgx_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kVec = 4;

// Store up to 4 consecutive floats starting at element e (of total n_elems).
__device__ __forceinline__ void store4(float *base, int64_t e, int64_t n_elems, bool aligned,
                                       const float v[4]) {
    if (aligned && e + 4 <= n_elems) {
        *reinterpret_cast<float4 *>(base + e) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (e + j < n_elems) base[e + j] = v[j];
    }
}

__device__ __forceinline__ void load4(const float *base, int64_t e, int64_t n_elems, bool aligned,
                                      float v[4]) {
    if (aligned && e + 4 <= n_elems) {
        float4 t = *reinterpret_cast<const float4 *>(base + e);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (e + j < n_elems) ? base[e + j] : 0.0f;
    }
}

__host__ __device__ inline bool is_aligned16(const void *p) {
    return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

__global__ void __launch_bounds__(kBlock) k_selftest_sincos(double *os, double *oc, const double *x, int64_t n) {
    int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        double sn, cs;
        pgx::pgx_sincos(x[i], sn, cs);
        os[i] = pgx::pgx_sin(x[i]);
        oc[i] = cs;
        if (sn != os[i]) oc[i] = __builtin_nan("");      // the two entry points must agree
    }
}

__global__ void __launch_bounds__(kBlock) k_selftest_tanh(double *out, const double *x, int64_t n) {
    int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[i] = pgx::pgx_tanh(x[i]);
}

// ------------------------------------------------------------------------------ fill / ramp / dirac
__global__ void __launch_bounds__(kBlock) k_fill(float *out, int64_t n_elems, float value, bool aligned) {
    int64_t stride = (int64_t)gridDim.x * kBlock * kVec;
    for (int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kVec; e < n_elems; e += stride) {
        float v[4] = {value, value, value, value};
        store4(out, e, n_elems, aligned, v);
    }
}

// mode 0: IdentityPE; mode 1: DiracPE (1 at absolute frame 0).
// IdentityPE is np.arange(start, start+n, dtype=float32) (identity_pe.py:54), which numpy fills as
// first + float(i) * delta with first = float32(start), delta = float32(start+1) - first, all in
// float32 -- for |start| >= 2^24 that is NOT float(start+i), and the quirk is reproduced here.
__global__ void __launch_bounds__(kBlock) k_index_source(float *out, int64_t start, int64_t n, int channels,
                                                         int mode, float first, float delta, bool aligned) {
    int64_t n_elems = n * channels;
    int64_t stride = (int64_t)gridDim.x * kBlock * kVec;
    for (int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kVec; e < n_elems; e += stride) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int64_t i = (e + j) / channels;
            v[j] = (mode == 0) ? (first + (float)i * delta) : ((start + i) == 0 ? 1.0f : 0.0f);
        }
        store4(out, e, n_elems, aligned, v);
    }
}

// The same fill for a WINDOW of consecutive blocks of `period` frames each (read_ahead.py): every block is the
// reference's own np.arange of that block -- first and delta from the block's start -- so that beyond 2^24, where
// the fill depends on where a block begins, a window hands out exactly the blocks the caller would have rendered.
__global__ void __launch_bounds__(kBlock) k_index_blocks(float *out, int64_t start, int64_t n, int channels,
                                                         int64_t period) {
    const int64_t n_elems = n * channels;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n_elems; e += stride) {
        const int64_t i = e / channels;
        const int64_t b = i / period;
        const int64_t sb = start + b * period;
        const float first = (float)sb;                              // round to nearest even, as numpy's cast
        const float delta = (float)(sb + 1) - first;
        out[e] = first + (float)(i - b * period) * delta;
    }
}

__global__ void __launch_bounds__(kBlock) k_window_copy(float *out, int64_t start, int64_t n, int channels,
                                                        const float *src, int64_t src_start, int64_t src_len,
                                                        int hold_first, int hold_last) {
    int64_t n_elems = n * channels;
    int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n_elems; e += stride) {
        int64_t f = e / channels;
        int c = (int)(e - f * channels);
        int64_t s = start + f - src_start;   // index into src
        float v = 0.0f;
        if (s >= 0 && s < src_len) v = src[s * channels + c];
        else if (s < 0 && hold_first) v = src[c];
        else if (s >= src_len && hold_last) v = src[(src_len - 1) * channels + c];
        out[e] = v;
    }
}

__global__ void __launch_bounds__(kBlock) k_extract_channel(float *out, const float *in, int64_t n, int channels,
                                                            int ch) {
    int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t f = (int64_t)blockIdx.x * kBlock + threadIdx.x; f < n; f += stride) out[f] = in[f * channels + ch];
}

// ------------------------------------------------------------------------------ SinePE (pure)
// sine_pe.py:159-175 + :141: phase = phase0 + w * (double(n)/sr); y = amp * sin(phase).
// post_gain: GainPE(SinePE, gain=<scalar>) fused -- the sine is rounded to float32 first and THEN
// multiplied by the float32 gain, i.e. the two roundings of the separate PEs (gain_pe.py:121-123).
__global__ void __launch_bounds__(kBlock) k_sine(float *out, int64_t out_stride, int64_t start, int64_t n,
                                                 int channels, double sr, const pgx_sine_params *params,
                                                 int has_gain, float post_gain) {
    const pgx_sine_params p = params[blockIdx.y];
    float *o = out + (int64_t)blockIdx.y * out_stride;
    const bool aligned = is_aligned16(o);
    const double inv_sr = 1.0 / sr;                              // t = n / sr below: pgx_div_by, same roundings
    int64_t n_elems = n * channels;
    int64_t stride = (int64_t)gridDim.x * kBlock * kVec;
    for (int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kVec; e < n_elems; e += stride) {
        float v[4];
        if (channels == 1) {
            // mono: four phases first; while they are all below the fast range of the sine (200 hours of a
            // 440 Hz tone) the four evaluations are one basic block and interleave -- same bits as pgx_sin
            double ph[4];
            bool fast = true;
            const double n0 = (double)(start + e);               // frame indices are exact in float64: one
#pragma unroll                                                   // conversion, then exact additions
            for (int j = 0; j < 4; ++j) {
                const double t = pgx::pgx_div_by(n0 + (double)j, sr, inv_sr);
                ph[j] = p.phase0 + p.w * t;
                fast = fast && (fabs(ph[j]) < pgx::kSinFastRange);
            }
            if (fast) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (float)(p.amp * pgx::pgx_sin_bounded(ph[j]));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (float)(p.amp * pgx::pgx_sin(ph[j]));
            }
            if (has_gain) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = v[j] * post_gain;
            }
            store4(o, e, n_elems, aligned, v);
            continue;
        }
        // frame of element e + j without a 64-bit division per element (software on this hardware): one
        // division per thread and a running remainder
        int64_t f = e;
        int rem = 0;
        if (channels != 1) {
            f = e / channels;
            rem = (int)(e - f * channels);
        }
        float cur = 0.0f;
        bool have = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!have) {
                double t = pgx::pgx_div_by((double)(start + f), sr, inv_sr);
                double ph = p.phase0 + p.w * t;
                cur = (float)(p.amp * pgx::pgx_sin(ph));
                if (has_gain) cur = cur * post_gain;
                have = true;
            }
            v[j] = cur;
            if (channels == 1 || ++rem == channels) {            // next element starts a new frame
                rem = 0;
                ++f;
                have = false;
            }
        }
        store4(o, e, n_elems, aligned, v);
    }
}

// ------------------------------------------------------------------------------ GainPE
__global__ void __launch_bounds__(kBlock) k_gain_const(float *out, const float *in, int64_t n_elems, float g,
                                                       bool aligned) {
    int64_t stride = (int64_t)gridDim.x * kBlock * kVec;
    for (int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kVec; e < n_elems; e += stride) {
        float v[4];
        load4(in, e, n_elems, aligned, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] * g;
        store4(out, e, n_elems, aligned, v);
    }
}

__global__ void __launch_bounds__(kBlock) k_gain_vec(float *out, const float *in, const float *gain, int64_t n,
                                                     int channels, int gain_channels, bool aligned) {
    int64_t n_elems = n * channels;
    int64_t stride = (int64_t)gridDim.x * kBlock * kVec;
    for (int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kVec; e < n_elems; e += stride) {
        float v[4], g[4];
        load4(in, e, n_elems, aligned, v);
        if (gain_channels == channels) {
            load4(gain, e, n_elems, aligned, g);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int64_t f = (e + j) / channels;
                g[j] = (f < n) ? gain[f] : 0.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] * g[j];
        store4(out, e, n_elems, aligned, v);
    }
}

// ------------------------------------------------------------------------------ MixPE
constexpr int kMixMax = 16;
struct MixPtrs {
    const float *p[kMixMax];
};

__global__ void __launch_bounds__(kBlock) k_mix_n(float *out, MixPtrs ins, int k, int accumulate,
                                                  int64_t n_elems, bool aligned) {
    int64_t stride = (int64_t)gridDim.x * kBlock * kVec;
    for (int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kVec; e < n_elems; e += stride) {
        float acc[4], v[4];
        int first = 0;
        if (accumulate) {
            load4(out, e, n_elems, aligned, acc);
        } else {
            load4(ins.p[0], e, n_elems, aligned, acc);
            first = 1;
        }
        for (int i = first; i < k; ++i) {
            load4(ins.p[i], e, n_elems, aligned, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = acc[j] + v[j];
        }
        store4(out, e, n_elems, aligned, acc);
    }
}

// VEC = 4: 16 B/lane (large mixes); VEC = 1: one float per lane so that a 48 000-frame block still
// spreads over ~190 workgroups.  The voice loop is unrolled 8x: the loads are independent of the
// (ordered, float32) additions, so 8 of them are in flight per lane.
template <int VEC>
__global__ void __launch_bounds__(kBlock) k_mix_batch(float *out, const float *in, int64_t in_stride, int batch,
                                                      int64_t n_elems, bool aligned) {
    int64_t stride = (int64_t)gridDim.x * kBlock * VEC;
    for (int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * VEC; e < n_elems; e += stride) {
        if (VEC == 4) {
            float acc[4], v[8][4];
            load4(in, e, n_elems, aligned, acc);
            int b = 1;
            for (; b + 8 <= batch; b += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) load4(in + (int64_t)(b + u) * in_stride, e, n_elems, aligned, v[u]);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = acc[j] + v[u][j];
            }
            for (; b < batch; ++b) {
                load4(in + (int64_t)b * in_stride, e, n_elems, aligned, v[0]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = acc[j] + v[0][j];
            }
            store4(out, e, n_elems, aligned, acc);
        } else {
            // one element per thread (blocks below 2^20 elements: a 48 000-frame block is 750 waves): 32 loads in
            // flight per lane, as k_gain_mix_batch keeps them, or the pass is latency-bound (512 inputs: 27 us)
            // (the chunk the batch ends in is loaded as a whole too, predicated: no tail of dependent round trips)
            float acc = in[e], v[32];
            int b = 1;
            for (; b + 32 <= batch; b += 32) {
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = in[(int64_t)(b + u) * in_stride + e];
#pragma unroll
                for (int u = 0; u < 32; ++u) acc = acc + v[u];
            }
            if (b < batch) {
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = (b + u < batch) ? in[(int64_t)(b + u) * in_stride + e] : 0.0f;
#pragma unroll
                for (int u = 0; u < 32; ++u)
                    if (b + u < batch) acc = acc + v[u];
            }
            out[e] = acc;
        }
    }
}

// out = sum_b float32(x_b * g_b): GainPE(voice, gain=<PE>) fused into the mix (gain_pe.py:104-119 then
// mix_pe.py:91-94).  Each product is rounded to float32 before the ordered float32 addition, exactly as
// when the two PEs run separately.  g is (frames, 1) or (frames, channels) per voice.
// (U voices of one frame: all 2U loads issued before the first product -- what keeps HBM busy is the number of loads in
// flight per thread.  PARTIAL: the batch ends inside the chunk; the loads stay unconditional in form (predicated), so
// a bank of 64 voices is two batched chunks, not one chunk and 31 dependent round trips: 16.5 -> 6 us.)
template <int U, bool FIRST, bool PARTIAL>
__device__ __forceinline__ void gain_mix_chunk(float &acc, const float *x, const float *g, int64_t x_stride,
                                               int64_t g_stride, int64_t e, int64_t ge, int b, int batch) {
    float xv[U], gv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool on = !PARTIAL || b + u < batch;
        xv[u] = on ? x[(int64_t)(b + u) * x_stride + e] : 0.0f;
        gv[u] = on ? g[(int64_t)(b + u) * g_stride + ge] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float prod = xv[u] * gv[u];
        if (FIRST && u == 0) acc = prod;
        else if (!PARTIAL || b + u < batch) acc = acc + prod;
    }
}

__global__ void __launch_bounds__(kBlock) k_gain_mix_batch(float *out, const float *x, const float *g,
                                                           int64_t x_stride, int64_t g_stride, int batch,
                                                           int64_t n, int channels, int gain_channels) {
    const int64_t n_elems = n * channels;
    int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n_elems; e += stride) {
        const int64_t ge = (gain_channels == channels) ? e : e / channels;
        // one thread per frame (the additions are ordered by voice), 32 voices x 2 streams in flight
        constexpr int U = 32;
        float acc;
        int b = 0;
        if (batch >= U) {
            gain_mix_chunk<U, true, false>(acc, x, g, x_stride, g_stride, e, ge, 0, batch);
            for (b = U; b + U <= batch; b += U)
                gain_mix_chunk<U, false, false>(acc, x, g, x_stride, g_stride, e, ge, b, batch);
            if (b < batch) gain_mix_chunk<U, false, true>(acc, x, g, x_stride, g_stride, e, ge, b, batch);
        } else {
            gain_mix_chunk<U, true, true>(acc, x, g, x_stride, g_stride, e, ge, 0, batch);
        }
        out[e] = acc;
    }
}

// ------------------------------------------------------------------------------ gates / triggers
// function_gen_pe.py:157-193 (pure, rectangle) + periodic_gate.py:63-67.
__global__ void __launch_bounds__(kBlock) k_periodic_gate(float *out, int64_t out_stride, int64_t start,
                                                          int64_t n, const pgx_gate_params *params) {
    const pgx_gate_params p = params[blockIdx.y];
    float *o = out + (int64_t)blockIdx.y * out_stride;
    const bool aligned = is_aligned16(o);
    int64_t stride = (int64_t)gridDim.x * kBlock * kVec;
    for (int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kVec; e < n; e += stride) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double idx = (double)(start + e + j);
            double base = pgx::pgx_mod1(idx * p.dt);
            double ph = pgx::pgx_mod1(base + p.phase);
            v[j] = (ph < p.duty) ? 1.0f : 0.0f;
        }
        store4(o, e, n, aligned, v);
    }
}

__global__ void __launch_bounds__(kBlock) k_periodic_trigger(float *out, int64_t start, int64_t n, int64_t period,
                                                             int64_t phase_samples, float amplitude, bool aligned) {
    int64_t stride = (int64_t)gridDim.x * kBlock * kVec;
    for (int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kVec; e < n; e += stride) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int64_t a = start + e + j + phase_samples;
            v[j] = (a % period == 0) ? amplitude : 0.0f;
        }
        store4(out, e, n, aligned, v);
    }
}

// ------------------------------------------------------------------------------ SuperSaw voice sum
// super_saw_pe.py:304-316: float64 accumulate of the float32 voices in voice order, * amp,
// -> float32, tiled over channels.
__global__ void __launch_bounds__(kBlock) k_supersaw_sum(float *out, int64_t out_stride, int nvoices, int64_t n,
                                                         int channels, const float *voices,
                                                         const double *amp_scalar, const float *amp,
                                                         int64_t amp_stride) {
    int b = blockIdx.y;
    float *o = out + (int64_t)b * out_stride;
    const float *vbase = voices + (int64_t)b * nvoices * n;
    const float *a = amp ? amp + (int64_t)b * amp_stride : nullptr;
    double as = amp_scalar ? amp_scalar[b] : 1.0;
    // four frames per thread: 16-byte loads of every voice when the rows are aligned (n % 4 == 0 keeps every voice
    // row 16-byte aligned; the buffers come from the pool, 256-byte aligned)
    const bool vec = (n % 4 == 0) && is_aligned16(vbase) && (!a || is_aligned16(a));
    int64_t stride = (int64_t)gridDim.x * kBlock * 4;
    for (int64_t f0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4; f0 < n; f0 += stride) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        if (vec) {
            for (int v = 0; v < nvoices; ++v) {
                const float4 x = *reinterpret_cast<const float4 *>(vbase + (int64_t)v * n + f0);
                acc[0] += (double)x.x; acc[1] += (double)x.y; acc[2] += (double)x.z; acc[3] += (double)x.w;
            }
        } else {
            for (int v = 0; v < nvoices; ++v)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (f0 + j < n) acc[j] += (double)vbase[(int64_t)v * n + f0 + j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t f = f0 + j;
            if (f >= n) break;
            double g = a ? (double)a[f] : as;
            float y = (float)(acc[j] * g);
            for (int c = 0; c < channels; ++c) o[f * channels + c] = y;
        }
    }
}

}  // namespace

// ================================================================================ C ABI
extern "C" {

int pgx_selftest_sincos(double *out_sin, double *out_cos, const double *x, int64_t n) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out_sin && out_cos && x, "pgx_selftest_sincos: null pointer");
    hipLaunchKernelGGL(k_selftest_sincos, dim3(pgx::grid_for(n, kBlock)), dim3(kBlock), 0, pgx::stream(), out_sin,
                       out_cos, x, n);
    PGX_LAUNCH_CHECK("k_selftest_sincos");
    return PGX_OK;
}

int pgx_selftest_tanh(double *out, const double *x, int64_t n) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && x, "pgx_selftest_tanh: null pointer");
    hipLaunchKernelGGL(k_selftest_tanh, dim3(pgx::grid_for(n, kBlock)), dim3(kBlock), 0, pgx::stream(), out, x, n);
    PGX_LAUNCH_CHECK("k_selftest_tanh");
    return PGX_OK;
}

int pgx_fill(float *out, int64_t n_elems, float value) {
    PGX_REQUIRE_INIT();
    if (n_elems <= 0) return PGX_OK;
    PGX_CHECK_ARG(out != nullptr, "pgx_fill: null output");
    hipLaunchKernelGGL(k_fill, dim3(pgx::grid_for(pgx::ceil_div(n_elems, kVec), kBlock)), dim3(kBlock), 0,
                       pgx::stream(), out, n_elems, value, is_aligned16(out));
    PGX_LAUNCH_CHECK("k_fill");
    return PGX_OK;
}

int pgx_ramp(float *out, float first, float delta, int64_t n, int channels) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out != nullptr && channels >= 1, "pgx_ramp: bad argument");
    hipLaunchKernelGGL(k_index_source, dim3(pgx::grid_for(pgx::ceil_div(n * channels, kVec), kBlock)),
                       dim3(kBlock), 0, pgx::stream(), out, (int64_t)0, n, channels, 0, first, delta,
                       is_aligned16(out));
    PGX_LAUNCH_CHECK("k_index_source");
    return PGX_OK;
}

int pgx_ramp_blocks(float *out, int64_t start, int64_t n, int channels, int64_t period) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out != nullptr && channels >= 1 && period >= 1, "pgx_ramp_blocks: bad argument");
    hipLaunchKernelGGL(k_index_blocks, dim3(pgx::grid_for(n * channels, kBlock)), dim3(kBlock), 0, pgx::stream(), out,
                       start, n, channels, period);
    PGX_LAUNCH_CHECK("k_index_blocks");
    return PGX_OK;
}

int pgx_dirac(float *out, int64_t start, int64_t n, int channels) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out != nullptr && channels >= 1, "pgx_dirac: bad argument");
    hipLaunchKernelGGL(k_index_source, dim3(pgx::grid_for(pgx::ceil_div(n * channels, kVec), kBlock)),
                       dim3(kBlock), 0, pgx::stream(), out, start, n, channels, 1, 0.0f, 0.0f,
                       is_aligned16(out));
    PGX_LAUNCH_CHECK("k_index_source");
    return PGX_OK;
}

int pgx_window_copy(float *out, int64_t start, int64_t n, int channels, const float *src, int64_t src_start,
                    int64_t src_len, int hold_first, int hold_last) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && src && channels >= 1 && src_len >= 1, "pgx_window_copy: bad argument");
    hipLaunchKernelGGL(k_window_copy, dim3(pgx::grid_for(n * channels, kBlock)), dim3(kBlock), 0, pgx::stream(),
                       out, start, n, channels, src, src_start, src_len, hold_first, hold_last);
    PGX_LAUNCH_CHECK("k_window_copy");
    return PGX_OK;
}

int pgx_extract_channel(float *out, const float *in, int64_t n, int channels, int ch) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && channels >= 1 && ch >= 0 && ch < channels, "pgx_extract_channel: bad argument");
    hipLaunchKernelGGL(k_extract_channel, dim3(pgx::grid_for(n, kBlock)), dim3(kBlock), 0, pgx::stream(), out, in,
                       n, channels, ch);
    PGX_LAUNCH_CHECK("k_extract_channel");
    return PGX_OK;
}

int pgx_sine_render(float *out, int64_t out_stride, int batch, int64_t start, int64_t n, int channels,
                    double sample_rate, const pgx_sine_params *params) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && params && channels >= 1 && sample_rate > 0, "pgx_sine_render: bad argument");
    PGX_CHECK_ARG(batch == 1 || out_stride >= n * channels, "pgx_sine_render: out_stride too small");
    PGX_CHECK_ARG(batch <= 65535, "pgx_sine_render: batch too large");
    dim3 grid(pgx::grid_for(pgx::ceil_div(n * channels, kVec), kBlock), batch);
    hipLaunchKernelGGL(k_sine, grid, dim3(kBlock), 0, pgx::stream(), out, out_stride, start, n, channels,
                       sample_rate, params, 0, 1.0f);
    PGX_LAUNCH_CHECK("k_sine");
    return PGX_OK;
}

int pgx_sine_gain_render(float *out, int64_t out_stride, int batch, int64_t start, int64_t n, int channels,
                         double sample_rate, const pgx_sine_params *params, float gain) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && params && channels >= 1 && sample_rate > 0, "pgx_sine_gain_render: bad argument");
    PGX_CHECK_ARG(batch == 1 || out_stride >= n * channels, "pgx_sine_gain_render: out_stride too small");
    PGX_CHECK_ARG(batch <= 65535, "pgx_sine_gain_render: batch too large");
    dim3 grid(pgx::grid_for(pgx::ceil_div(n * channels, kVec), kBlock), batch);
    hipLaunchKernelGGL(k_sine, grid, dim3(kBlock), 0, pgx::stream(), out, out_stride, start, n, channels,
                       sample_rate, params, 1, gain);
    PGX_LAUNCH_CHECK("k_sine<gain>");
    return PGX_OK;
}

int pgx_gain_const(float *out, const float *in, int64_t n_elems, float gain) {
    PGX_REQUIRE_INIT();
    if (n_elems <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in, "pgx_gain_const: null pointer");
    bool al = is_aligned16(out) && is_aligned16(in);
    hipLaunchKernelGGL(k_gain_const, dim3(pgx::grid_for(pgx::ceil_div(n_elems, kVec), kBlock)), dim3(kBlock), 0,
                       pgx::stream(), out, in, n_elems, gain, al);
    PGX_LAUNCH_CHECK("k_gain_const");
    return PGX_OK;
}

int pgx_gain_vec(float *out, const float *in, const float *gain, int64_t n, int channels, int gain_channels) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && gain && channels >= 1, "pgx_gain_vec: bad argument");
    PGX_CHECK_ARG(gain_channels == 1 || gain_channels == channels,
                  "pgx_gain_vec: gain must be mono or match the source channel count");
    bool al = is_aligned16(out) && is_aligned16(in) && is_aligned16(gain);
    hipLaunchKernelGGL(k_gain_vec, dim3(pgx::grid_for(pgx::ceil_div(n * channels, kVec), kBlock)), dim3(kBlock), 0,
                       pgx::stream(), out, in, gain, n, channels, gain_channels, al);
    PGX_LAUNCH_CHECK("k_gain_vec");
    return PGX_OK;
}

int pgx_mix_n(float *out, const float *const *ins_host, int k, int64_t n_elems) {
    PGX_REQUIRE_INIT();
    if (n_elems <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && ins_host && k >= 1, "pgx_mix_n: bad argument");
    bool al = is_aligned16(out);
    for (int i = 0; i < k; ++i) {
        PGX_CHECK_ARG(ins_host[i] != nullptr, "pgx_mix_n: null input");
        al = al && is_aligned16(ins_host[i]);
    }
    int done = 0;
    while (done < k) {
        MixPtrs ptrs;
        int cnt = (k - done < kMixMax) ? (k - done) : kMixMax;
        for (int i = 0; i < kMixMax; ++i) ptrs.p[i] = (i < cnt) ? ins_host[done + i] : nullptr;
        hipLaunchKernelGGL(k_mix_n, dim3(pgx::grid_for(pgx::ceil_div(n_elems, kVec), kBlock)), dim3(kBlock), 0,
                           pgx::stream(), out, ptrs, cnt, done > 0 ? 1 : 0, n_elems, al);
        PGX_LAUNCH_CHECK("k_mix_n");
        done += cnt;
    }
    return PGX_OK;
}

int pgx_mix_batch(float *out, const float *in, int64_t in_stride, int batch, int64_t n_elems) {
    PGX_REQUIRE_INIT();
    if (n_elems <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && batch >= 1 && in_stride >= n_elems, "pgx_mix_batch: bad argument");
    if (n_elems >= (int64_t)1 << 20) {
        bool al = is_aligned16(out) && is_aligned16(in) && (in_stride % 4 == 0);
        hipLaunchKernelGGL(k_mix_batch<4>, dim3(pgx::grid_for(pgx::ceil_div(n_elems, 4), kBlock)), dim3(kBlock),
                           0, pgx::stream(), out, in, in_stride, batch, n_elems, al);
    } else {
        hipLaunchKernelGGL(k_mix_batch<1>, dim3(pgx::grid_for(n_elems, kBlock)), dim3(kBlock), 0, pgx::stream(),
                           out, in, in_stride, batch, n_elems, false);
    }
    PGX_LAUNCH_CHECK("k_mix_batch");
    return PGX_OK;
}

int pgx_gain_mix_batch(float *out, const float *in, int64_t in_stride, const float *gain, int64_t gain_stride,
                       int batch, int64_t n, int channels, int gain_channels) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && gain && batch >= 1 && channels >= 1, "pgx_gain_mix_batch: bad argument");
    PGX_CHECK_ARG(gain_channels == 1 || gain_channels == channels,
                  "pgx_gain_mix_batch: gain must be mono or match the source channel count");
    PGX_CHECK_ARG(in_stride >= n * channels && gain_stride >= n * gain_channels,
                  "pgx_gain_mix_batch: stride too small");
    hipLaunchKernelGGL(k_gain_mix_batch, dim3(pgx::grid_for(n * channels, kBlock)), dim3(kBlock), 0,
                       pgx::stream(), out, in, gain, in_stride, gain_stride, batch, n, channels, gain_channels);
    PGX_LAUNCH_CHECK("k_gain_mix_batch");
    return PGX_OK;
}

int pgx_periodic_gate(float *out, int64_t out_stride, int batch, int64_t start, int64_t n,
                      const pgx_gate_params *params) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && params, "pgx_periodic_gate: null pointer");
    PGX_CHECK_ARG(batch == 1 || out_stride >= n, "pgx_periodic_gate: out_stride too small");
    PGX_CHECK_ARG(batch <= 65535, "pgx_periodic_gate: batch too large");
    dim3 grid(pgx::grid_for(pgx::ceil_div(n, kVec), kBlock), batch);
    hipLaunchKernelGGL(k_periodic_gate, grid, dim3(kBlock), 0, pgx::stream(), out, out_stride, start, n, params);
    PGX_LAUNCH_CHECK("k_periodic_gate");
    return PGX_OK;
}

int pgx_periodic_trigger(float *out, int64_t start, int64_t n, int64_t period, int64_t phase_samples,
                         float amplitude) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && period > 0, "pgx_periodic_trigger: bad argument");
    hipLaunchKernelGGL(k_periodic_trigger, dim3(pgx::grid_for(pgx::ceil_div(n, kVec), kBlock)), dim3(kBlock), 0,
                       pgx::stream(), out, start, n, period, phase_samples, amplitude, is_aligned16(out));
    PGX_LAUNCH_CHECK("k_periodic_trigger");
    return PGX_OK;
}

int pgx_supersaw_sum(float *out, int64_t out_stride, int batch, int nvoices, int64_t n, int channels,
                     const float *voices, const double *amp_scalar, const float *amp, int64_t amp_stride) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && voices && nvoices >= 1 && channels >= 1, "pgx_supersaw_sum: bad argument");
    PGX_CHECK_ARG(batch == 1 || out_stride >= n * channels, "pgx_supersaw_sum: out_stride too small");
    PGX_CHECK_ARG(batch <= 65535, "pgx_supersaw_sum: batch too large");
    dim3 grid(pgx::grid_for(pgx::ceil_div(n, 4), kBlock), batch);
    hipLaunchKernelGGL(k_supersaw_sum, grid, dim3(kBlock), 0, pgx::stream(), out, out_stride, nvoices, n,
                       channels, voices, amp_scalar, amp, amp_stride);
    PGX_LAUNCH_CHECK("k_supersaw_sum");
    return PGX_OK;
}

}  // extern "C"
