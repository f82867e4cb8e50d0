// pgx_seq.hip -- PEs whose recurrences are not associative scans: LadderPE (nonlinear) and
// CombPE (integer-indexed delay line).  (The ADSR state machines live in pgx_adsr.hip.)
//
// Parallelism comes from independent chains (voices x channels); inside one chain the
// reference's per-sample operation order is followed literally (-ffp-contract=off).

#include "pgx_common.h"

namespace {

// ================================================================================================
// LadderPE (ladder_pe.py:31-203).  One lane per (instance, channel) chain.
// ================================================================================================
__global__ void __launch_bounds__(64)
k_ladder(float *out, int64_t out_stride, const float *in, int64_t in_stride, int batch, int64_t n, int channels,
         double sr, const pgx_ladder_params *params, const float *freq, const float *resonance,
         const float *drive, double *state) {
    const int chain = blockIdx.x * 64 + threadIdx.x;
    if (chain >= batch * channels) return;
    const int inst = chain / channels, ch = chain - inst * channels;
    const pgx_ladder_params p = params[inst];
    const float *x = in + (int64_t)inst * in_stride;
    float *o = out + (int64_t)inst * out_stride;
    double *st = state + (int64_t)chain * 9;
    double z0[4] = {st[0], st[1], st[2], st[3]};
    double z1[4] = {st[4], st[5], st[6], st[7]};
    double old_input = st[8];

    const int oversample = p.oversample;
    const double oversample_recip = 1.0 / (double)oversample;
    const double state_decay = 0.95, input_threshold = 1e-5, resonance_multiplier = 1.8;
    const double two_pi = 2.0 * 3.141592653589793;
    const double min_cutoff = 5.0;
    const double nyquist = sr / 2.0;
    double max_cutoff = nyquist * 0.85;
    if (max_cutoff > nyquist - 1.0) max_cutoff = nyquist - 1.0;
    const double pbg = p.passband_gain;
    const int mode = p.mode;

    for (int64_t i = 0; i < n; ++i) {
        double cutoff = freq ? (double)freq[i] : p.freq;
        if (cutoff < min_cutoff) cutoff = min_cutoff;
        if (cutoff > max_cutoff) cutoff = max_cutoff;
        const double wc = cutoff * two_pi / (sr * (double)oversample);
        const double wc2 = wc * wc;
        const double wc3 = wc2 * wc;
        const double wc4 = wc3 * wc;
        const double alpha = 0.9892 * wc - 0.4324 * wc2 + 0.1381 * wc3 - 0.0202 * wc4;
        const double q_adjust = 1.006 + 0.0536 * wc - 0.095 * wc2 - 0.05 * wc4;

        double res = resonance ? (double)resonance[i] : p.resonance;
        if (res < 0.0) res = 0.0;
        if (res > 1.0) res = 1.0;
        const double k = 4.0 * res * resonance_multiplier;

        double drv = drive ? (double)drive[i] : p.drive;
        double drive_scaled;
        if (drv < 0.0) drv = 0.0;
        if (drv > 1.0) {
            if (drv > 4.0) drv = 4.0;
            drive_scaled = 1.0 + (drv - 1.0) * (1.0 - pbg);
        } else {
            drive_scaled = drv;
        }

        const double input_sample = (double)x[i * channels + ch] * drive_scaled;
        const double input_abs = input_sample >= 0.0 ? input_sample : -input_sample;
        if (input_abs < input_threshold) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                z0[s] *= state_decay;
                z1[s] *= state_decay;
            }
            old_input *= state_decay;
        }
        double total = 0.0, interp = 0.0;
        for (int os = 0; os < oversample; ++os) {
            const double in_interp = interp * old_input + (1.0 - interp) * input_sample;
            const double u = tanh(in_interp - (z1[3] - pbg * in_interp) * k * q_adjust);
            double ft, stage1, stage2, stage3, stage4, weighted;

            ft = u * 0.76923077 + 0.23076923 * z0[0] - z1[0];
            ft = ft * alpha + z1[0];
            z1[0] = ft; z0[0] = u; stage1 = ft;

            ft = stage1 * 0.76923077 + 0.23076923 * z0[1] - z1[1];
            ft = ft * alpha + z1[1];
            z1[1] = ft; z0[1] = stage1; stage2 = ft;

            ft = stage2 * 0.76923077 + 0.23076923 * z0[2] - z1[2];
            ft = ft * alpha + z1[2];
            z1[2] = ft; z0[2] = stage2; stage3 = ft;

            ft = stage3 * 0.76923077 + 0.23076923 * z0[3] - z1[3];
            ft = ft * alpha + z1[3];
            z1[3] = ft; z0[3] = stage3; stage4 = ft;

            if (mode == 0) weighted = stage4;
            else if (mode == 1) weighted = stage2;
            else if (mode == 2) weighted = (stage2 + stage4) * 4.0 - stage3 * 8.0;
            else if (mode == 3) weighted = (stage1 - stage2) * 2.0;
            else if (mode == 4) weighted = u + stage4 - (stage1 + stage3) * 4.0 + stage2 * 6.0;
            else weighted = u + stage2 - stage1 * 2.0;

            total += weighted * oversample_recip;
            interp += oversample_recip;
        }
        old_input = input_sample;
        o[i * channels + ch] = (float)total;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        st[s] = z0[s];
        st[4 + s] = z1[s];
    }
    st[8] = old_input;
}

// ================================================================================================
// CombPE (comb_pe.py:26-113).
// Kernel 1 (one lane): the control recurrence -- smoothed frequency -> integer delay, clamped
// feedback -- is inherently sequential and must be exact because it produces INDICES.
// Kernel 2 (one workgroup per channel): y[n] = x[n] + fb[n]*ring[n - D[n]]; samples whose delay
// reaches before the current chunk start are independent, so chunks of up to 256 such samples are
// processed in parallel (reads, barrier, writes, barrier).
// ================================================================================================
__global__ void __launch_bounds__(256)
k_comb_control(int64_t n, double sr, double freq_scalar, double fb_scalar, const float *freq, const float *fb,
               double min_frequency, double smooth_alpha, int64_t buffer_len, double max_feedback,
               double *state, int32_t *delay, double *fbv) {
    // feedback clamp: parallel
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        double f = fb ? (double)fb[i] : fb_scalar;
        if (!isfinite(f)) f = 0.0;
        if (f > max_feedback) f = max_feedback;
        if (f < -max_feedback) f = -max_feedback;
        fbv[i] = f;
    }
    __shared__ long long s_fill_from;
    __shared__ int s_fill_val;
    if (threadIdx.x == 0) {
        double smoothed = state[1];
        long long fill_from = n;
        int fill_val = 1;
        const bool constant = (freq == nullptr);
        const double raw_c = freq_scalar < min_frequency ? min_frequency : freq_scalar;
        for (int64_t i = 0; i < n; ++i) {
            double raw = raw_c;
            if (!constant) {
                raw = (double)freq[i];
                if (raw < min_frequency) raw = min_frequency;
            }
            const double prev = smoothed;
            if (smoothed < 0.0) smoothed = raw;
            else smoothed += (raw - smoothed) * smooth_alpha;
            double f = smoothed < 1.0 ? 1.0 : smoothed;
            int64_t d = (int64_t)rint(sr / f);
            if (d < 1) d = 1;
            if (d >= buffer_len) d = buffer_len - 1;
            delay[i] = (int32_t)d;
            // constant frequency: once the one-pole no longer moves, every later sample repeats
            // this delay exactly -- hand the rest to the parallel fill below.
            if (constant && smoothed == prev) {
                fill_from = i + 1;
                fill_val = (int)d;
                break;
            }
        }
        state[1] = smoothed;
        // advance write_pos here (kernel 2 derives its start position from the new value)
        int64_t wp = (int64_t)state[0];
        wp = (wp + n) % buffer_len;
        state[0] = (double)wp;
        s_fill_from = fill_from;
        s_fill_val = fill_val;
    }
    __syncthreads();
    for (int64_t i = s_fill_from + threadIdx.x; i < n; i += blockDim.x) delay[i] = s_fill_val;
}

__global__ void __launch_bounds__(256)
k_comb_apply(float *out, const float *in, int64_t n, int channels, double *ring, int64_t buffer_len,
             const double *state, const int32_t *delay, const double *fbv) {
    __shared__ int s_first_bad[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = blockIdx.x;
    const int64_t wp_new = (int64_t)state[0];
    int64_t wp = ((wp_new - n) % buffer_len + buffer_len) % buffer_len;   // write position at block start
    int64_t pos = 0;
    while (pos < n) {
        // chunk length S: first j with pos+j >= n or delay[pos+j] <= j
        const int64_t i = pos + tid;
        int32_t d = 0;
        bool ok = false;
        if (i < n) {
            d = delay[i];
            ok = d > tid;
        }
        unsigned long long bad = __ballot(!ok);
        int fb_lane = bad ? (__ffsll((long long)bad) - 1) : 64;
        if (lane == 0) s_first_bad[wave] = fb_lane;
        __syncthreads();
        int S = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            int v = s_first_bad[w];
            if (S == w * 64) S += v;          // extend only while all previous waves were fully ok
        }
        // S >= 1 always (delay >= 1 > 0 for tid 0 when pos < n)
        double y = 0.0;
        if (tid < S) {
            int64_t rp = (wp + tid - d) % buffer_len;
            if (rp < 0) rp += buffer_len;
            const double delayed = ring[rp * channels + ch];
            y = (double)in[i * channels + ch] + fbv[i] * delayed;
            out[i * channels + ch] = (float)y;
        }
        __syncthreads();
        if (tid < S) ring[((wp + tid) % buffer_len) * channels + ch] = y;
        __syncthreads();
        wp = (wp + S) % buffer_len;
        pos += S;
    }
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

int pgx_ladder(float *out, int64_t out_stride, const float *in, int64_t in_stride, int batch, int64_t n,
               int channels, double sample_rate, const pgx_ladder_params *params, const float *freq,
               const float *resonance, const float *drive, double *state) {
    PGX_REQUIRE_INIT();
    if (n <= 0 || batch <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && params && state && channels >= 1 && sample_rate > 0, "pgx_ladder: bad argument");
    PGX_CHECK_ARG(batch == 1 || (out_stride >= n * channels && in_stride >= n * channels),
                  "pgx_ladder: instance stride too small");
    PGX_CHECK_ARG(batch == 1 || (!freq && !resonance && !drive),
                  "pgx_ladder: per-sample control streams require batch == 1");
    int chains = batch * channels;
    hipLaunchKernelGGL(k_ladder, dim3((chains + 63) / 64), dim3(64), 0, pgx::stream(), out, out_stride, in,
                       in_stride, batch, n, channels, sample_rate, params, freq, resonance, drive, state);
    PGX_LAUNCH_CHECK("k_ladder");
    return PGX_OK;
}

int pgx_comb(float *out, const float *in, int64_t n, int channels, double sample_rate, double freq_scalar,
             double fb_scalar, const float *freq, const float *fb, double min_frequency,
             int64_t smoothing_samples, double *ring, int64_t buffer_len, double *state, int32_t *delay_scratch,
             double *fb_scratch) {
    PGX_REQUIRE_INIT();
    if (n <= 0) return PGX_OK;
    PGX_CHECK_ARG(out && in && ring && state && delay_scratch && fb_scratch, "pgx_comb: null pointer");
    PGX_CHECK_ARG(channels >= 1 && buffer_len >= 2 && smoothing_samples >= 1 && sample_rate > 0,
                  "pgx_comb: bad argument");
    hipLaunchKernelGGL(k_comb_control, dim3(1), dim3(256), 0, pgx::stream(), n, sample_rate, freq_scalar,
                       fb_scalar, freq, fb, min_frequency, 1.0 / (double)smoothing_samples, buffer_len, 0.995,
                       state, delay_scratch, fb_scratch);
    PGX_LAUNCH_CHECK("k_comb_control");
    hipLaunchKernelGGL(k_comb_apply, dim3(channels), dim3(256), 0, pgx::stream(), out, in, n, channels, ring,
                       buffer_len, (const double *)state, (const int32_t *)delay_scratch,
                       (const double *)fb_scratch);
    PGX_LAUNCH_CHECK("k_comb_apply");
    return PGX_OK;
}

}  // extern "C"
