/*
 * pygmu_hip.h -- C ABI of libpygmu_hip.so, the MI355X (gfx950) render library behind
 * pygmu2's `ProcessingElement._render(start, duration) -> Snippet` hot path.
 *
 * The reference (rdpoor/pygmu2) is pure Python/numpy: it has no FFI.  The drop-in
 * boundary is therefore the body of each PE's `_render`; every entry
 * point below replaces the numpy/scipy/numba arithmetic of one reference function,
 * cited as `file:line` relative to /root/reference/src/pygmu2/.  The Python binding a
 * maintainer would add is shown in INTEGRATION.md (ctypes; pygmu2_amd/device.py is
 * that binding for this repository's own PE classes).
 *
 * Conventions
 *  - Plain C types only.  `float*` / `double*` / `void*` arguments are DEVICE pointers
 *    obtained from pgx_malloc unless the name ends in `_host`.
 *  - Audio buffers are (frames, channels) row-major float32, exactly the reference's
 *    Snippet payload (snippet.py:37-45).  Batched entry points take `batch` independent
 *    instances ("voices"); instance i's buffer starts `*_stride` ELEMENTS after
 *    instance i-1's, its parameter block is params[i], its state is state[i].
 *  - Parameter blocks and state blobs live in device memory and are owned by the caller
 *    (the PE object).  A zero-filled state blob is the reset state unless noted.
 *  - Every call is asynchronous on the library stream (pgx_stream_handle) unless
 *    documented as synchronous.  No entry point allocates or synchronises unless noted,
 *    so sequences of calls can be captured into a HIP graph.
 *  - Return value: 0 on success, negative pgx_status otherwise; the message of the last
 *    failure on the calling thread is available from pgx_last_error().  No C++
 *    exception crosses this boundary.
 */
#ifndef PYGMU_HIP_H
#define PYGMU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PGX_OK = 0,
    PGX_ERR_INVALID = -1,   /* bad argument (maps to Python ValueError)            */
    PGX_ERR_RUNTIME = -2,   /* HIP runtime failure (maps to Python RuntimeError)   */
    PGX_ERR_NOT_INIT = -3,  /* pgx_init not called / no device                      */
    PGX_ERR_NOMEM = -4
} pgx_status;

/* ------------------------------------------------------------------ runtime */
int pgx_abi_version(void);
const char *pgx_last_error(void);
int pgx_device_count(int *count);
int pgx_init(int device);                 /* select device, create stream + pool; idempotent */
int pgx_shutdown(void);                   /* sync, release pool, destroy stream              */
int pgx_device_name(char *buf, size_t len);
void *pgx_stream_handle(void);            /* hipStream_t of the library stream               */
int pgx_stream_sync(void);
/* Two independent sub-graphs of one block may overlap on the device:
 *   pgx_stream_fork()      a side stream starts behind everything enqueued so far; calls go to it
 *   pgx_stream_select(s)   while forked: s != 0 -> side stream, 0 -> main stream
 *   pgx_stream_join()      main stream waits for the side stream; calls go to the main stream again
 * Between fork and join pgx_free() parks blocks (they are re-pooled at the join): "freed = reusable"
 * is only true in single-stream order.  The caller keeps the two call sequences data-independent. */
int pgx_stream_fork(void);
int pgx_stream_select(int side);
int pgx_stream_join(void);
int pgx_stream_is_forked(void);           /* 1 between fork and join */
/* A fork whose side work the main stream does not need yet (the next block's envelopes, rendered one block ahead):
 *   pgx_stream_detach()         ends the fork without waiting: the side stream keeps running what it was given, calls go
 *                               to the main stream again; blocks freed during the fork stay parked
 *   pgx_stream_wait_detached()  the main stream waits for that work (no-op when nothing is detached)
 * A new fork waits for detached work first (there is one side stream); pgx_stream_sync covers it. */
int pgx_stream_detach(void);
/* pgx_stream_fork_after(event): a fork whose side stream starts behind `event` -- recorded earlier on the main stream
 * with pgx_event_record -- instead of behind the main stream's current tail (NULL: behind nothing).  For side work that
 * depends on nothing the main stream is doing and writes only buffers whose last main-stream reader the event covers. */
int pgx_stream_fork_after(void *event);
int pgx_stream_wait_detached(void);
int pgx_stream_is_detached(void);

int pgx_malloc(void **dptr, size_t bytes);      /* pooled (size-class free lists)           */
int pgx_free(void *dptr);                       /* returns the block to the pool            */
int pgx_pool_trim(void);                        /* hipFree every cached block               */
int pgx_memset(void *dptr, int byte_value, size_t bytes);
int pgx_memcpy_h2d(void *dst, const void *src_host, size_t bytes);  /* synchronous */
int pgx_memcpy_d2h(void *dst_host, const void *src, size_t bytes);  /* synchronous */
int pgx_memcpy_d2d(void *dst, const void *src, size_t bytes);

/* Root Snippets leaving the device (the caller of render() reads `.data`: benchmark_pes.py:176-185 hands host
 * arrays to its caller).  Pinned host blocks are pooled like device blocks.
 *   pgx_d2h_begin   asynchronous copy on the library's copy stream, ordered behind everything enqueued on
 *                   the library stream so far -- the next block renders while this one crosses PCIe
 *   pgx_d2h_wait    the host blocks until that copy has landed
 *   pgx_d2h_fence   the library stream waits for it instead (before `src` is recycled unread) */
int pgx_host_malloc(void **hptr, size_t bytes);
int pgx_host_free(void *hptr);
int pgx_d2h_begin(void *dst_host, const void *src, size_t bytes, int64_t *ticket);
int pgx_d2h_wait(int64_t ticket);
int pgx_d2h_query(int64_t ticket, int *done);   /* *done = 1 once that copy has landed; never blocks */
int pgx_d2h_fence(int64_t ticket);

int pgx_event_create(void **event);
int pgx_event_destroy(void *event);
int pgx_event_record(void *event);              /* on the library stream */
int pgx_event_elapsed_ms(void *start, void *stop, float *ms);   /* synchronises on stop */

/* Diagnostics: evaluate the library's float64 sine / cosine (the routine every oscillator and
 * coefficient kernel uses in place of np.sin / np.cos) on n device doubles. */
int pgx_selftest_sincos(double *out_sin, double *out_cos, const double *x, int64_t n);
/* the library's own float64 tanh (LadderPE's feedback nonlinearity), for the accuracy test */
int pgx_selftest_tanh(double *out, const double *x, int64_t n);

/* ------------------------------------------------------------------ sources / copies
 * ConstantPE._render (constant_pe.py:51-63), IdentityPE._render (identity_pe.py:43-60),
 * DiracPE._render (dirac_pe.py:48-67), ArrayPE._render (array_pe.py:74-129) and the
 * copy/hold part of _ExtentWindowPE._render (extent_window_pe.py:88-157).  Bit-exact.
 */
int pgx_fill(float *out, int64_t n_elems, float value);
/* IdentityPE: out[i, :] = first + float(i) * delta in float32, where the host passes
 * first = float32(start), delta = float32(start + 1) - first (numpy's arange fill rule). */
int pgx_ramp(float *out, float first, float delta, int64_t n, int channels);
/* IdentityPE over a window of consecutive blocks of `period` frames (n frames from `start`): every block is the
 * reference's np.arange(block start, ..., dtype=float32) -- first and delta taken at the block's own start, which is
 * what the fill depends on beyond 2^24 (identity_pe.py:54). */
int pgx_ramp_blocks(float *out, int64_t start, int64_t n, int channels, int64_t period);
int pgx_dirac(float *out, int64_t start, int64_t n, int channels);
/* out[f, :] = src[start + f - src_start, :] where start+f lies in [src_start, src_start+src_len),
 * else src[0] if (before && hold_first), src[src_len-1] if (after && hold_last), else 0. */
int pgx_window_copy(float *out, int64_t start, int64_t n, int channels,
                    const float *src, int64_t src_start, int64_t src_len,
                    int hold_first, int hold_last);

/* out[f] = in[f, ch]: channel selection of a control stream, as _scalar_or_pe_values does
 * for a multi-channel parameter PE (processing_element.py:346-352). */
int pgx_extract_channel(float *out, const float *in, int64_t n, int channels, int ch);

/* ------------------------------------------------------------------ SinePE
 * Pure path: SinePE._render + _compute_phase_pure (sine_pe.py:119-175).
 *   phase = phase0 + w * (double(n) / sr),  out = float(amp * sin(phase)),  w = (2*pi)*f
 *   computed on the host exactly as the reference does.  float64 throughout.
 */
typedef struct {
    double w;        /* (2.0 * pi) * frequency */
    double amp;
    double phase0;
} pgx_sine_params;
int pgx_sine_render(float *out, int64_t out_stride, int batch, int64_t start, int64_t n,
                    int channels, double sample_rate, const pgx_sine_params *params);
/* GainPE(SinePE(scalars), gain=<scalar>) in one launch: float32(float32(amp*sin) * float32 gain),
 * the same two roundings as pgx_sine_render followed by pgx_gain_const (gain_pe.py:121-123). */
int pgx_sine_gain_render(float *out, int64_t out_stride, int batch, int64_t start, int64_t n,
                         int channels, double sample_rate, const pgx_sine_params *params, float gain);

/* Stateful path: _compute_phase_stateful (sine_pe.py:177-232).  freq/amp/phase_mod are
 * optional per-sample float32 control streams (frames, 1); NULL -> the scalar in params.
 * state[instance] = { accumulated_phase, initialised flag }.  The phase term is added to
 * every sample and is included in the carried phase, as in the reference. */
typedef struct {
    double freq;
    double amp;
    double phase;          /* scalar phase (used as initial phase and as per-sample term) */
    int32_t phase_is_stream;
    int32_t pad;
} pgx_sine_stateful_params;
int pgx_sine_stateful(float *out, int64_t n, int channels, double sample_rate,
                      const pgx_sine_stateful_params *params,
                      const float *freq, const float *amp, const float *phase_mod,
                      double *state /* [2] */);

/* ------------------------------------------------------------------ GainPE / MixPE
 * GainPE._render (gain_pe.py:92-127): one float32 multiply -> bit-exact.
 * MixPE._render (mix_pe.py:91-94): float32 adds in input order -> bit-exact. */
int pgx_gain_const(float *out, const float *in, int64_t n_elems, float gain);
int pgx_gain_vec(float *out, const float *in, const float *gain, int64_t n, int channels,
                 int gain_channels /* 1 (broadcast) or == channels */);
/* ins_host: HOST array of k device pointers, each n_elems floats. */
int pgx_mix_n(float *out, const float *const *ins_host, int k, int64_t n_elems);
/* Sum of `batch` equally shaped voices stored [batch][n_elems] (stride in elements),
 * accumulated in float32 in voice order 0..batch-1 (== MixPE over the voices). */
int pgx_mix_batch(float *out, const float *in, int64_t in_stride, int batch, int64_t n_elems);
/* MixPE over voices that each end in GainPE(voice, gain=<PE>): out = sum_b float32(in_b * gain_b),
 * products rounded to float32, added in voice order -- identical to pgx_gain_vec + pgx_mix_batch. */
int pgx_gain_mix_batch(float *out, const float *in, int64_t in_stride, const float *gain,
                       int64_t gain_stride, int batch, int64_t n, int channels, int gain_channels);

/* ------------------------------------------------------------------ BiquadPE
 * Constant coefficients: BiquadPE._filter_constant_coeffs (biquad_pe.py:383-404), i.e.
 * scipy.signal.lfilter's direct-form-II-transposed section in float64:
 *     y = z0 + b0*x;  z0 = (z1 + b1*x) - a1*y;  z1 = b2*x - a2*y
 * evaluated by a parallel scan over affine state maps; each thread re-runs its chunk in
 * exactly this operation order from its scanned carry-in.
 * coef[instance] = {b0,b1,b2,a1,a2} (host computes them, biquad_pe.py:217-335);
 * state[instance][channel] = {z0,z1}.  workspace: device scratch of at least
 * pgx_biquad_workspace_bytes(batch, n, channels, settle_frames) bytes (may be NULL when that is 0).
 * settle_frames: 0, or a frame count W for which every entry of A^W (A = [[-a1,1],[-a2,0]], all
 * instances) is below 2^-90, evaluated by the host from the coefficients it computed.  With W > 0 a
 * long chain is rendered in one launch: each workgroup rebuilds its carry-in from the W frames before
 * its range, which is exact to far below one float64 ulp; W = 0 (or a W above 65536) uses the
 * reduce + apply launch pair, exact for any section.
 * tables: NULL, or [batch][pgx_biquad_table_doubles()] doubles filled once per coefficient set by
 * pgx_biquad_tables (the powers of A the scan uses); saves the single-launch path its prologue. */
size_t pgx_biquad_workspace_bytes(int batch, int64_t n, int channels, int64_t settle_frames);
size_t pgx_biquad_table_doubles(void);
int pgx_biquad_tables(double *tables, const double *coef /* [batch][5] */, int batch);
int pgx_biquad_const(float *out, int64_t out_stride, const float *in, int64_t in_stride,
                     int batch, int64_t n, int channels,
                     const double *coef /* [batch][5] */, const double *tables, int64_t settle_frames,
                     double *state /* [batch][channels][2] */, void *workspace);

/* BiquadPE(SinePE) with scalar parameters, mono (biquad_pe.py:383-404 over sine_pe.py:159-175): the sine is made
 * in registers inside the single-launch filter kernel, so the chain writes 4 B per frame and reads nothing.
 * w = (2 pi) f, amp, phase0 as in pgx_sine_params; start = first frame of the render.  Only where
 * pgx_biquad_sine_supported(n, settle_frames) (the settled single-launch plan applies) and the phase stays below
 * 2e9 rad; the caller renders the two PEs separately otherwise.  A thread's first frame is k_sine's sample bit for
 * bit, its next 15 are that (sin, cos) pair turned by w / sr: within the rounding noise of the reference's phase. */
int pgx_biquad_sine_supported(int64_t n, int64_t settle_frames);
int pgx_biquad_sine(float *out, int64_t start, int64_t n, double sample_rate, double w, double amp, double phase0,
                    const double *coef /* [5] */, const double *tables, int64_t settle_frames,
                    double *state /* [2] */,
                    double *state_backup /* [2] or NULL: receives the state on entry (a window's snapshot) */);

/* Time-varying coefficients: _compute_coefficients per sample (biquad_pe.py:217-335) +
 * the direct-form-I recurrence of _biquad_varying_numba (biquad_pe.py:35-62).
 * freq/q: per-sample float32 control streams (frames,1) or NULL -> scalar.
 * state[channel] = {x1,x2,y1,y2}.  mode: 0 lowpass,1 highpass,2 bandpass,3 notch,
 * 4 allpass,5 peaking,6 lowshelf,7 highshelf. */
typedef struct {
    double freq;
    double q;
    double gain_db;
    int32_t mode;
    int32_t pad;
} pgx_biquad_var_params;
/* gain_a = 10^(gain_db/40) and gain_sqrt_a = sqrt(gain_a), computed by the host exactly as the
 * reference does (biquad_pe.py:250,290).
 * workspace: pgx_scan2_workspace_bytes(n, channels) bytes, or NULL.  With a workspace a long block is cut
 * into segments: a reduce launch composes each segment's affine state map, the apply launch folds the
 * earlier maps onto the carried state and renders (2 launches x up to 128 workgroups per channel);
 * without one (or for blocks of one or two tiles) a single workgroup walks the block. */
size_t pgx_scan2_workspace_bytes(int64_t n, int channels);
int pgx_biquad_varying(float *out, const float *in, int64_t n, int channels, double sample_rate,
                       const pgx_biquad_var_params *params, const float *freq, const float *q,
                       double gain_a, double gain_sqrt_a, double *state /* [channels][4] */,
                       void *workspace);

/* ------------------------------------------------------------------ SVFilterPE / EnvelopePE / TransformPE
 * (the PEs of benchmarks/profile_biquad_vs_svfilter.py)
 * SVFilterPE: _svf_coefficients_batch_numba + _svf_varying_numba (svfilter_pe.py:65-205).
 * params reuses pgx_biquad_var_params; mode: 0 lowpass,1 highpass,2 bandpass,3 notch,4 peaking,
 * 5 lowshelf,6 highshelf.  gain_a = 10^(gain_db/40).  state[channel] = {s0, s1}. */
int pgx_svf(float *out, const float *in, int64_t n, int channels, double sample_rate,
            const pgx_biquad_var_params *params, const float *freq, const float *q, double gain_a,
            const double *coef /* NULL, or {a00,a01,a10,a11,b0,b1,c0,c1,c2} evaluated by the host
                                  (constant frequency and q; freq and q must then be NULL) */,
            double *state /* [channels][2] */, void *workspace /* as for pgx_biquad_varying */);

/* EnvelopePE._render (envelope_pe.py:128-206) on the (already look-ahead shifted) source block.
 * one_pole != 0: attack == release, scipy lfilter one-pole as a scan; else the attack/release
 * switch of _envelope_ar_numba (envelope_pe.py:259-271), one lane per channel.
 * rms_window > 0 selects DetectionMode.RMS (block-local uniform_filter1d, mode='nearest'); rms_period > 0: the
 * n frames are several of the caller's blocks of rms_period frames rendered at once, and the detector restarts
 * at every one of their edges (0: one block).
 * state[channel] = envelope; scratch: pgx_envelope_scratch_bytes (detector output + 64-frame block sums, and
 * the pieces / entries of the all-windows form used from 131 072 frames on). */
size_t pgx_envelope_scratch_bytes(int64_t n, int channels);
int pgx_envelope(float *out, const float *in, int64_t n, int channels, double attack_coeff,
                 double release_coeff, int one_pole, int rms_window, int64_t rms_period, double *state,
                 double *scratch);

/* TransformPE._render (transform_pe.py:96-152) for chains of named element-wise operations:
 * float32 -> float64 -> ops in order -> float32.
 * code: 0 affine (p1 + p0*x), 1 clip [p0,p1], 2 sqrt, 3 square, 4 abs, 5 tanh, 6 one_minus. */
typedef struct {
    int32_t code;
    int32_t pad;
    double p0;
    double p1;
} pgx_transform_op;
int pgx_transform(float *out, const float *in, int64_t n_elems, const pgx_transform_op *ops, int nops);

/* ------------------------------------------------------------------ DelayPE / PiecewisePE / WAV formats
 * (the callers and data formats either side of the path)
 * DelayPE's float / PE delay: interpolated_lookup (interpolated_lookup.py:28-130) over the rendered
 * source window [window_start, window_start + window_len): index = float64(start + i) - delay,
 * linear or Catmull-Rom ("cubic"), neighbours clipped to the window; with bounded != 0 indices
 * outside [extent_start, extent_end) give 0 (delay_pe.py:199-205).  delay: NULL -> delay_scalar. */
int pgx_interp_lookup(float *out, const float *window, int64_t window_start, int64_t window_len,
                      int channels, int64_t start, int64_t n, double delay_scalar, const float *delay,
                      int cubic, int bounded, double extent_start, double extent_end);
/* result_dev[0..1] = min, max of float64(start + i) - delay[i]  (np.min / np.max of the indices,
 * interpolated_lookup.py:111-112): the host sizes the source window from them. */
int pgx_index_range(double *result_dev, const float *delay, int64_t start, int64_t n);
/* partials_dev[2 p + {0, 1}] = (min, max) of workgroup p's share of the mono stream x[0..n), p < parts <= 1024
 * (NaN if it met one): the host folds the pairs.  LadderPE with a PE-driven cutoff / resonance sizes the warm-up
 * of its time segments from the block's lowest cutoff and highest resonance (ladder_pe.py:84-113 clamps). */
int pgx_stream_range(double *partials_dev, int parts, const float *x, int64_t n);
/* PiecewisePE._render (piecewise_pe.py:164-229); times sorted int64, values float64 (device).
 * transition: 0 step, 1 linear, 2 exponential, 3 sigmoid, 4 constant_power. */
int pgx_piecewise(float *out, int64_t start, int64_t n, int channels, const int64_t *times,
                  const double *values, int count, int transition, int hold_first, int hold_last);
/* WavWriterPE / WavReaderPE sample conversion on the device (half the PCIe bytes of float32):
 * libsndfile's PCM_16 rules as python-soundfile configures them (clipping on; the reference's default
 * subtype, wav_writer_pe.py:67): s = x * 32768 saturated to [-32768, 32767], else lrintf; x = s / 32768. */
int pgx_f32_to_pcm16(int16_t *out, const float *in, int64_t n_elems);
int pgx_pcm16_to_f32(float *out, const int16_t *in, int64_t n_elems);

/* ------------------------------------------------------------------ SpatialPE (section 8f rank 2)
 * SpatialAdapter.render (spatial_pe.py:94-144): M -> N channels. */
int pgx_channel_adapt(float *out, const float *in, int64_t n, int src_channels, int out_channels);
/* SpatialLinear / SpatialConstantPower.render (spatial_pe.py:179-214, 250-286): mono mix of the source
 * panned to stereo; azimuth in degrees, clipped to +-90; azimuth_stream: per-frame float32 or NULL. */
int pgx_pan(float *out, const float *in, int64_t n, int src_channels, float azimuth,
            const float *azimuth_stream, int constant_power);
/* out[i] = float32 mean of frame i's channels (the mono mix SpatialHRTF convolves, spatial_pe.py:483) */
int pgx_mono_mean(float *out, const float *in, int64_t n, int src_channels);

/* ------------------------------------------------------------------ LoopPE / WindowPE / DynamicsPE
 * (the remaining configurations of benchmarks/benchmark_pes.py:309-350)
 * LoopPE._render (loop_pe.py:159-232): out[i] = loop[(start + i) mod loop_len] (numpy's non-negative modulo),
 * silence from total_len on (count * loop_len; < 0 = endless); the last `crossfade` frames of the loop blend
 * into its first ones with the reference's float64 weights.  `loop` = the source rendered over the loop region
 * (loop_len, channels).  Bit-exact. */
int pgx_loop(float *out, const float *loop, int64_t start, int64_t n, int channels, int64_t loop_len,
             int64_t total_len, int64_t crossfade);
/* WindowPE._render (window_pe.py:118-254): centred window of 2*half_window + 1 frames; `padded` = the source
 * rendered over [start - half_window, start + n + half_window).  mode 0 max, 1 min, 2 mean, 3 rms;
 * rectify: |x| first.  max / min are exact, mean / rms sum the window directly (the reference differences a
 * cumulative sum).  workspace: pgx_window_workspace_bytes (64-frame block statistics, float64). */
size_t pgx_window_workspace_bytes(int64_t n, int channels, int64_t half_window);
int pgx_window(float *out, const float *padded, int64_t n, int channels, int64_t half_window, int mode,
               int rectify, void *workspace);
/* DynamicsPE._render (dynamics_pe.py:190-372): out = audio * 10**((gain_db(20 log10(max(env, 1e-10))) +
 * makeup) / 20) with numpy's float32 typing of every step.  The host rounds the Python scalars the way numpy's
 * weak-scalar promotion does (pygmu2_amd/dynamics_pe.py). */
typedef struct {
    int mode;              /* 0 compress, 1 limit (ratio = inf), 2 expand, 3 gate */
    int soft;              /* knee > 0 */
    int stereo_link;
    int wide_makeup;       /* the make-up gain is a numpy float64 scalar (automatic value): float64 from its addition on */
    float threshold;       /* float32(threshold) */
    float slope;           /* float32(1/ratio - 1) (compress), float32(ratio - 1) (expand) */
    float neg_slope;       /* float32(-(ratio - 1)) */
    float half_knee;       /* float32(knee / 2) */
    float two_knee;        /* float32(2 * knee) */
    float knee;            /* float32(knee) */
    float knee_lo;         /* float32(threshold - knee/2) */
    float knee_hi;         /* float32(threshold + knee/2) */
    float gate_range;      /* float32(gate_range) */
    float makeup;          /* float32(makeup_gain_db) */
    double gate_range_d;   /* the hard-knee gate is float64 in numpy */
    double makeup_d;
} pgx_dynamics_params;
int pgx_dynamics(float *out, const float *audio, const float *envelope, int64_t n, int channels,
                 int env_channels, const pgx_dynamics_params *params);

/* ------------------------------------------------------------------ BlitSawPE / SuperSawPE
 * BlitSawPE._render (blit_saw_pe.py:150-264): phase cumsum -> mod 1 -> Dirichlet kernel
 * -> leaky integrator -> *2 *amp -> float32.  state[instance] = {phase, integrator}.
 * The caller applies the reset-on-discontinuity rule (blit_saw_pe.py:182-185) by
 * re-initialising state to {initial_phase, 0} before the call.
 * freq/amp/m streams: optional per-sample float32 (frames,1), stride in elements between
 * instances (0 = shared). */
typedef struct {
    double freq;
    double amp;
    double leak;
    double m;              /* <= 0: automatic (largest odd M below Nyquist) */
} pgx_blitsaw_params;
int pgx_blitsaw(float *out, int64_t out_stride, int batch, int64_t n, int channels,
                double sample_rate, const pgx_blitsaw_params *params,
                const float *freq, int64_t freq_stride,
                const float *amp, int64_t amp_stride,
                const float *m, int64_t m_stride,
                double *state /* [batch][2] */,
                void *workspace /* pgx_blitsaw_workspace_bytes, or NULL: one workgroup per instance */,
                double *state_backup /* NULL, or [batch][2]: receives the state on entry -- the snapshot of a caller
                                        that renders a block ahead of its stream and may have to take it back */);
/* A long stream of a few scalar-parameter oscillators is rendered by several workgroups per oscillator
 * (two passes that replay the single workgroup's carry chains: same bits).  0 = not applicable. */
size_t pgx_blitsaw_workspace_bytes(int batch, int64_t n, int streams /* any of freq/amp/m is a stream */);

/* SuperSawPE._render (super_saw_pe.py:287-318): out = float(amp * sum_v double(voice_v))
 * where voice_v is the float32 output of BlitSaw v.  `voices` = [batch*nvoices] float32
 * buffers laid out [instance][voice][frames]; amp stream optional. */
int pgx_supersaw_sum(float *out, int64_t out_stride, int batch, int nvoices, int64_t n,
                     int channels, const float *voices, const double *amp_scalar /* [batch] */,
                     const float *amp, int64_t amp_stride);

/* A bank of mono BiquadPE(BlitSawPE) voices with scalar parameters in one launch: the oscillator's float32
 * samples go through the constant-coefficient section (scipy's DF-II-T order, pgx_biquad_const) without leaving
 * the chip.  params / saw_state as for pgx_blitsaw; coef = [batch][5] {b0,b1,b2,a1,a2}; biquad_state = [batch][2].
 * One workgroup per voice: for banks of >= 128 voices. */
int pgx_blitsaw_biquad_bank(float *out, int64_t out_stride, int batch, int64_t n, double sample_rate,
                            const pgx_blitsaw_params *params, double *saw_state /* [batch][2] */,
                            const double *coef, double *biquad_state /* [batch][2] */);

/* The same chain with sixteen frames per thread: pgx_supersaw_wide's oscillator (under its conditions, which the
 * CALLER checks; saw_tables from pgx_supersaw_wide_tables with one voice per instance) and
 * the settled biquad's filter section on the tables of pgx_biquad_tables.  Agrees with pgx_blitsaw_biquad_bank to a
 * float32 ulp or two (<= 1e-6 of peak asserted), not to the bit. */
int pgx_blitsaw_biquad_wide(float *out, int64_t out_stride, int batch, int64_t n,
                            const double *saw_tables /* [batch] pgx_supersaw_wide_tables(nvoices = 1) */,
                            double *saw_state /* [batch][2] */, const double *coef /* [batch][5] */,
                            const double *biquad_tables /* [batch] pgx_biquad_tables */,
                            double *biquad_state /* [batch][2] */,
                            const float *gain /* NULL, or [batch][gain_stride]: out = float32(voice * gain), GainPE's product */,
                            int64_t gain_stride);
/* The same chain in concurrent time segments for banks of up to 256 voices (a rank's share of a sharded mix), so that a
 * few voices fill the chip with ONE launch: workgroup (voice, s) renders the 4096-frame tiles of segment s.  On entering
 * a segment the oscillator's phase is a product and its integrator level a closed form (the harmonics' steady state plus
 * the decayed remainder of the carried level); the filter has no closed form but forgets: the segment starts
 * ceil(settle_frames / 4096) tiles early from a zero filter state and emits nothing there (settle_frames: the caller's
 * bound for "every entry of A^W below 2^-90", the largest of the bank, > 0).  States are read from the *_in buffers and
 * written to the *_out buffers (two different buffers each).  <= 1e-6 of peak from pgx_blitsaw_biquad_wide. */
int pgx_blitsaw_biquad_wide_segments(int batch, int64_t n, int64_t settle_frames);
int pgx_blitsaw_biquad_wide_seg(float *out, int64_t out_stride, int batch, int64_t n, const double *saw_tables,
                                const double *saw_state_in, double *saw_state_out, const double *coef,
                                const double *biquad_tables, const double *biquad_state_in, double *biquad_state_out,
                                const float *gain, int64_t gain_stride, int64_t settle_frames);

/* MixPE over a bank of such voices -- MixPE(*[GainPE(BiquadPE(BlitSawPE), gain=<envelope>)]) or without the GainPE
 * (mix_pe.py:91-94 over gain_pe.py:104-119, biquad_pe.py:383-404, blit_saw_pe.py:150-264) -- MIXED ON CHIP: out[n] is the
 * mix's block, the [voices][frames] layer between the voices and the mix is never written.  The work is cut into
 * independent (voice, 4096-frame tile) pairs: a workgroup renders a group of voices over the same tile and adds them in a
 * float64 accumulator per frame; the groups' rows of partial sums are then added in group order (a fixed order).  A pair
 * enters with the oscillator's phase (a product), its integrator level (closed form, as pgx_blitsaw_biquad_wide_seg) and a
 * filter started from rest warm_frames before the first frame it emits (warm_frames: a multiple of 16 with every entry of
 * A^warm_frames below 2^-90 for every voice, <= pgx_voice_tiles_max_warm()).  tables: pgx_voice_tiles_tables packs each
 * voice's constants (saw_tables / coef / biquad_tables as for pgx_blitsaw_biquad_wide, plus the per-thread turns of the
 * oscillator's anchor) into one 12 KB block that a workgroup brings into LDS by LDS-DMA, one voice ahead of its use
 * (pgx_voice_tiles_table_bytes(nvoices) bytes).  States are read from the *_in buffers and written to the *_out buffers.
 * The oscillator's sample enters the filter unrounded (BlitSawPE's float32 rounding: 6e-8 of a voice's level); each voice's
 * float32 sample and its float32 product with the gain are the two-PE chain's; the sum is taken in float64 and rounded once (the reference adds float32 in voice
 * order): <= 1e-6 of peak from pgx_blitsaw_biquad_wide + pgx_gain_mix_batch. */
int64_t pgx_voice_tiles_max_warm(void);
size_t pgx_voice_tiles_table_bytes(int nvoices);
int pgx_voice_tiles_tables(double *tables, const double *saw_tables, const double *coef /* [nvoices][5] */,
                           const double *biquad_tables, int nvoices);
size_t pgx_voice_tiles_workspace_bytes(int nvoices, int64_t n, int64_t warm_frames);
/* What the tiles of a block enter with (integrator levels' closed-form terms, first anchors) depends on the oscillators'
 * phases only: pgx_voice_tiles_entries makes it for the block that begins advance_frames after the block saw_state is the
 * start of -- 0: that block itself; n: the next block of a stream, while this one is still being rendered (another
 * stream) -- into set `slot` (0 / 1) of the workspace; pgx_voice_tiles then takes entries_slot = that set, or -1 to make
 * them itself first. */
int pgx_voice_tiles_entries(void *workspace, int slot, int nvoices, int64_t n, const double *tables,
                            const double *saw_state /* [nvoices][2] */, int64_t advance_frames, int64_t warm_frames);
int pgx_voice_tiles(float *out /* [n] */, int nvoices, int64_t n, const double *tables,
                    const double *saw_state_in, double *saw_state_out /* [nvoices][2] */,
                    const double *biquad_state_in, double *biquad_state_out /* [nvoices][2] */,
                    const float *gain /* NULL, or [nvoices][gain_stride] */, int64_t gain_stride, int64_t warm_frames,
                    void *workspace /* pgx_voice_tiles_workspace_bytes(nvoices, n, warm_frames) */, int entries_slot,
                    int next_entries_slot /* -1, or the other set: the entries of the block that follows this one in the
                                             stream are made in the launch that adds the rows (from saw_state_out) */);

/* A bank of scalar-parameter SuperSawPEs in one launch, voices summed on chip: the same samples as
 * pgx_blitsaw over batch*nvoices oscillators followed by pgx_supersaw_sum, bit for bit, without the
 * [batch*nvoices][frames] intermediate (one workgroup per instance, one wave per oscillator; nvoices <= 16).
 * params / state are [batch*nvoices] as for pgx_blitsaw, laid out [instance][voice]. */
int pgx_supersaw_bank(float *out, int64_t out_stride, int batch, int nvoices, int64_t n, int channels,
                      double sample_rate, const pgx_blitsaw_params *params, double *state /* [batch*nvoices][2] */,
                      const double *amp_scalar /* [batch] */);
/* The same bank when a few instances have to fill the chip (a rank's share of a sharded mix): every instance is cut
 * into pgx_supersaw_bank_segments(batch, n) time segments rendered by concurrent workgroups of one launch.  A later
 * segment takes its oscillators' phase sums by replaying the per-tile additions and their integrator levels from
 * the closed form of the leaky integrator's response to the BLIT's harmonics (blit_saw_pe.py:196-235:
 * y = yss(phase) + leak^frames * (y0 - yss(phase0))); needs scalar frequencies, the automatic (odd) M and
 * leak < 1.  Agreement with the sequential recurrence ~1e-14 (float64), i.e. the same float32 samples up to a
 * rounding flip in ~1e-7 of them.  state_in is read, state_out (another buffer) receives the new states. */
int pgx_supersaw_bank_segments(int batch, int64_t n);
int pgx_supersaw_bank_seg(float *out, int64_t out_stride, int batch, int nvoices, int64_t n, int channels,
                          double sample_rate, const pgx_blitsaw_params *params,
                          const double *state_in, double *state_out, const double *amp_scalar /* [batch] */,
                          const double *tables /* pgx_supersaw_bank_tables, or NULL: made by every workgroup */);
/* What the bank's workgroups need per voice and what depends on the parameters only (Dirichlet constants, rotation
 * sines, powers of the leak, per-thread prefix offsets, per-lane scan powers): made once per bank, loaded by every
 * later launch.  The same operations as the in-kernel preparation, hence the same samples. */
size_t pgx_supersaw_bank_table_bytes(int batch, int nvoices);
int pgx_supersaw_bank_tables(double *tables, int batch, int nvoices, double sample_rate,
                             const pgx_blitsaw_params *params);
/* The bank with sixteen frames per thread: the per-thread part of the work (anchor sines, the voice's constants,
 * the integrator scan) is paid half as often -- the bank is bound by instruction issue.  Scalar frequencies >= 1 Hz
 * with the automatic (odd) M and |sin(M pi f/sr)| >= 0.05 only (always true below Nyquist): the CALLER checks that
 * (the rotation / recurrence form of the Dirichlet kernel is the only one here).  Frame phases are frac(phase0 + (i+1) * inc), not k_blitsaw's running sums: the output
 * agrees with pgx_supersaw_bank to ~1e-9 of peak, not to the bit.  Time segments (pgx_supersaw_wide_segments) as in
 * pgx_supersaw_bank_seg; tables from pgx_supersaw_wide_tables (their own layout); state_in != state_out. */
size_t pgx_supersaw_wide_table_bytes(int batch, int nvoices);
int pgx_supersaw_wide_tables(double *tables, int batch, int nvoices, double sample_rate,
                             const pgx_blitsaw_params *params);
int pgx_supersaw_wide_segments(int batch, int nvoices, int64_t n);
int pgx_supersaw_wide(float *out, int64_t out_stride, int batch, int nvoices, int64_t n, int channels,
                      const double *state_in, double *state_out, const double *amp_scalar /* [batch] */,
                      const double *tables);

/* ------------------------------------------------------------------ LadderPE
 * _ladder_process_numba (ladder_pe.py:31-203), the reference's float64 operation order.
 * state[instance][channel] = {z0[4], z1[4], old_input}.
 * settle_frames = 0: one lane per (instance, channel) chain, strictly sequential.
 * settle_frames = W > 0 (host estimate of how many samples the filter needs to forget its state: from the
 * small-signal loop below self-oscillation, a TRIAL value at and above it -- a driven, saturating ladder usually
 * locks to its input although the linearised loop does not decay; the caller reads the counters below and moves to
 * a longer warm-up or to W = 0 when chains fall back): the block is cut into segments that each start W samples
 * early from a zero state; the library verifies every segment against its left neighbour's final
 * state (1e-8 relative) and REPAIRS what disagrees -- the segment behind a bad boundary is rendered again from its
 * neighbour's exit state, sample by sample in the reference's operation order, and so are its successors until the
 * repaired trajectory meets a segment's own entry state again (a chain that disagrees from its first boundary on is
 * the sequential render, bit for bit) -- so W affects speed, never results beyond that bound.  accurate_frames = A (0 < A < W): only the last A warm-up samples
 * of a segment evaluate tanh in float64; the W - A before them, which merely have to forget the zero start,
 * use the float32 exponential unit (state error ~1e-7, contracted by the accurate tail; 0 or >= W: all of the
 * warm-up is accurate).  Warm-ups run on fused multiply-adds; emitted samples keep the reference's operation
 * order.  workspace: pgx_ladder_workspace_bytes(...) bytes
 * (NULL when 0).  Its first 16 bytes are counters, cumulative since the caller zeroed them: int32[0] chains with a
 * repair, int32[1] segmented launches, int64[1] samples repaired (the position does not depend on n). */
typedef struct {
    double freq;
    double resonance;
    double drive;
    double passband_gain;
    int32_t oversample;
    int32_t mode;          /* 0 lp24,1 lp12,2 bp24,3 bp12,4 hp24,5 hp12 */
} pgx_ladder_params;
int pgx_ladder(float *out, int64_t out_stride, const float *in, int64_t in_stride,
               int batch, int64_t n, int channels, double sample_rate,
               const pgx_ladder_params *params,
               const float *freq, const float *resonance, const float *drive, /* batch==1 only */
               double *state /* [batch][channels][9] */, int64_t settle_frames, int64_t accurate_frames,
               void *workspace);
size_t pgx_ladder_workspace_bytes(int batch, int64_t n, int channels, int64_t settle_frames);

/* ------------------------------------------------------------------ CombPE
 * _comb_process_numba (comb_pe.py:26-113): y[n] = x[n] + fb[n] * y[n - D[n]].
 * ring = [batch][2][ring_rows][channels] float64, zero at reset: the last buffer_len outputs, double buffered --
 * a render reads half `parity` and leaves the other half current; the write position on entering a render is
 * total_frames mod buffer_len (total_frames = frames rendered since the reset, kept by the caller).
 *  - scalar frequency (freq == NULL; any batch): params[v].delay = clip(rint(sr / max(max(f, min_f), 1)), 1,
 *    buffer_len - 1) (comb_pe.py:70-77 on the settled smoother), delay_min / delay_max = its range over the batch.
 *    fb (batch == 1) optionally streams the feedback.  Up to 1024 * delay frames the render is the reference's loop
 *    operation for operation; beyond, time segments run concurrently (carries through a float64 affine fold).
 *  - freq != NULL (batch == 1): the smoothed frequency is carried in state[0] (-1 = unset); delays come from a
 *    time-parallel evaluation of the one-pole (comb_pe.py:61-77), the ring runs in LDS.  ring_rows = buffer_len.
 *    The delays are index work: a sample whose sr / f lies within 1e-9 of a rounding tie (where the time-parallel
 *    level and the reference's chain, ~1e-14 apart, could round differently) raises a flag, and the block's delays
 *    are then made again by the reference's loop itself on one lane -- always the reference's integers.
 *    state = double[4]: {smoothed frequency, level on entering the last block, tie flag (int32), blocks redone by the
 *    literal loop (int64)}; the caller sets {-1, 0, 0, 0} at reset. */
typedef struct {
    double feedback;       /* scalar feedback (clamped to +-0.995 by the kernels, non-finite -> 0) */
    int32_t delay;         /* scalar-frequency delay in frames (unused with a frequency stream) */
    int32_t buffer_len;    /* ceil(sr / min_frequency) + 1 rows (comb_pe.py:216-218) */
} pgx_comb_params;
int pgx_comb(float *out, int64_t out_stride, const float *in, int64_t in_stride, int batch, int64_t n,
             int channels, double sample_rate, const pgx_comb_params *params /* device [batch] */,
             int delay_min, int delay_max, const float *freq, const float *fb /* batch == 1 */,
             double min_frequency, int64_t smoothing_samples,
             double *ring, int64_t ring_rows, int64_t total_frames, int parity,
             double *state /* [4], frequency stream only */, void *workspace);
size_t pgx_comb_workspace_bytes(int batch, int64_t n, int channels, int delay_max, int freq_stream);

/* ------------------------------------------------------------------ envelopes / gates
 * PeriodicGate (periodic_gate.py:63-67 over function_gen_pe.py:157-193, scalar params):
 *   gate = (mod(mod(double(n)*dt,1)+phase,1) < duty) ? 1 : 0.  Bit-exact.
 * PeriodicTrigger (periodic_trigger.py:49-58): amp where (n+phase_samples) % period == 0. */
typedef struct {
    double dt;             /* frequency / sample_rate */
    double phase;
    double duty;           /* already clipped to [0,1] */
} pgx_gate_params;
int pgx_periodic_gate(float *out, int64_t out_stride, int batch, int64_t start, int64_t n,
                      const pgx_gate_params *params);
int pgx_periodic_trigger(float *out, int64_t start, int64_t n, int64_t period,
                         int64_t phase_samples, float amplitude);
/* PeriodicGate with PE-driven parameters: FunctionGenPE's stateful rectangle path (function_gen_pe.py:169-186).
 * A stream pointer overrides its scalar; state = { carried phase in cycles } (the host zeroes it when a
 * render is not contiguous with the previous one, function_gen_pe.py:170-171). */
int pgx_gate_stateful(float *out, int64_t n, double sample_rate, double freq, double duty, double phase,
                      const float *freq_stream, const float *duty_stream, const float *phase_stream,
                      double *state);

/* AdsrGatedPE._render (adsr_pe.py:124-196) / AdsrTriggeredPE._render (adsr_pe.py:279-335):
 * the reference's sequential float64 accumulation reproduced bit for bit (binade-linear runs, see
 * pgx_adsr.hip).  state[instance] = {state enum, env, prev_gate | sustain_ends_at}.
 * workspace: >= pgx_adsr_workspace_bytes(batch, n) bytes of device scratch (edge masks).
 * A lone gated envelope (batch == 1) over 196 608 frames or more -- a look-ahead window -- is walked with its 65 536-frame
 * chunks as one batch, in rounds (every chunk from the carried state, then from its left neighbour's exit; settled when
 * two rounds' exits agree bit for bit -- every completed attack pins the state, so two rounds settle any envelope
 * whose chunks each hold one); this one case WAITS for the device once per call to read the verdict, and a render
 * that has not settled after three rounds is walked chunk after chunk.  Same samples either way. */
typedef struct {
    double attack_dvdt;
    double decay_dvdt;
    double release_dvdt;
    double sustain_level;
    int64_t sustain_samples;   /* triggered variant only */
} pgx_adsr_params;
size_t pgx_adsr_workspace_bytes(int batch, int64_t n);
int pgx_adsr_gated(float *out, int64_t out_stride, const float *gate, int64_t gate_stride,
                   int batch, int64_t n, const pgx_adsr_params *params, double *state, void *workspace);
/* AdsrGatedPE(PeriodicGate(scalar params)): the gate of pgx_periodic_gate is evaluated inside the
 * envelope kernels, sample for sample the same values, without materialising the gate buffer.
 * detach_walk != 0: after the (parallel) edge search the library forks (pgx_stream_fork), enqueues the
 * envelope walk -- one latency-bound wave per envelope -- on the side stream and selects the main
 * stream again; the caller enqueues independent work and must call pgx_stream_join() before any call
 * that reads `out` or touches `state`.  Not allowed while already forked. */
int pgx_adsr_gated_periodic(float *out, int64_t out_stride, int batch, int64_t start, int64_t n,
                            const pgx_gate_params *gates, const pgx_adsr_params *params, double *state,
                            void *workspace, int detach_walk);
/* The same block with the carried state read from one buffer and written to another (a block rendered ahead of the
 * caller's stream -- voice_bank.py: the states it started from stay where they are, a pull that turns out not to be
 * this block costs nothing to undo).  detach_walk = 0: everything on the current stream; != 0: as above -- the edge search on
 * the current stream, then the fork and the walk on the side stream (the caller joins or detaches). */
int pgx_adsr_gated_periodic_to(float *out, int64_t out_stride, int batch, int64_t start, int64_t n,
                               const pgx_gate_params *gates, const pgx_adsr_params *params,
                               const double *state_in, double *state_out, void *workspace, int detach_walk);
int pgx_adsr_triggered(float *out, int64_t out_stride, const float *trig, int64_t trig_stride,
                       int batch, int64_t start, int64_t n, const pgx_adsr_params *params,
                       double *state, void *workspace);

/* ------------------------------------------------------------------ ConvolvePE
 * ConvolvePE._render (convolve_pe.py:250-342): y = x * h (linear, streaming).  The
 * reference evaluates it by float64 FFT overlap-save; this library evaluates the same sum
 * directly as a Toeplitz(h) x Hankel(x) matrix product on the f32 MFMA units with
 * float64 accumulation across K-slabs.
 *   x:     (n, src_ch) float32 current block
 *   hist:  (L-1, out_ch) float32 previous inputs (fanned out), updated in place
 *   h:     (L, fir_ch) float32
 *   out:   (n, out_ch) float32
 * workspace >= pgx_convolve_workspace_bytes(n, L, out_ch). */
size_t pgx_convolve_workspace_bytes(int64_t n, int64_t fir_len, int out_channels);
int pgx_convolve(float *out, const float *x, int64_t n, int src_channels,
                 const float *h, int64_t fir_len, int fir_channels, int out_channels,
                 float *hist, void *workspace);

/* Long filters: the same convolution by float64 FFT overlap-save, hand-written (four-step N1 x N2
 * decomposition, Stockham radix-4 FFTs of 4096 points per workgroup in LDS, spectrum kept in the
 * order the forward pass produces it, two real blocks packed per complex transform).
 *   fft_size = pgx_convolve_fft_size(L): power of two >= 2L in [4096, 262144], 0 if L is too long
 *   spectrum: pgx_convolve_fft_spectrum_bytes(...) bytes, filled once per filter by _prepare
 *   workspace >= pgx_convolve_fft_workspace_bytes(n, L, out_ch, fft_size)
 * x, hist, out as for pgx_convolve; hist is updated in place. */
int64_t pgx_convolve_fft_size(int64_t fir_len);
size_t pgx_convolve_fft_spectrum_bytes(int64_t fft_size, int fir_channels);
size_t pgx_convolve_fft_workspace_bytes(int64_t n, int64_t fir_len, int out_channels, int64_t fft_size);
int pgx_convolve_fft_prepare(void *spectrum, const float *h, int64_t fir_len, int fir_channels,
                             int64_t fft_size);
int pgx_convolve_fft(float *out, const float *x, int64_t n, int src_channels, const void *spectrum,
                     int64_t fir_len, int fir_channels, int out_channels, int64_t fft_size,
                     float *hist, void *workspace,
                     int hist_is_zero /* fresh stream: the history counts as zeros and is only written */);

/* ------------------------------------------------------------------ multi-GPU exchange (RCCL over xGMI)
 * The one exchange step of the path: the partial mixes of a MixPE whose inputs are dealt i mod world over
 * the ranks are summed (mix_pe.py:91-94 is the sum being distributed).  One process
 * per GPU.  librccl is loaded on demand by these entry points only.
 *   pgx_comm_unique_id   rank 0 creates the 128-byte rendezvous id; the host hands it to every rank
 *                        (any channel: a file, MPI, torch.distributed's store ...)
 *   pgx_comm_init        collective: every rank calls it with the same id, after pgx_init(device)
 *   pgx_allreduce_sum    out[i] = sum over ranks of in[i] (float32, n elements; out may equal in).  Runs on
 *                        the library's collective stream, ordered behind everything enqueued on the library
 *                        stream so far; asynchronous.  `in` and `out` must stay allocated until the ticket
 *                        has been waited for.  The call records the ordering event and queues the job; a thread
 *                        of the library's own hands the jobs to RCCL in call order (PGX_COMM_THREAD=0: the
 *                        caller does, 16-18 us of host time per call).  Calls must come from one thread.
 *   pgx_allreduce_wait   orders the library stream behind the collective of `ticket` (stream-level wait:
 *                        the host blocks only until that collective has been handed to RCCL -- normally long
 *                        ago), after which `out` may be read and `in` released
 *   pgx_allreduce_scalar_host   synchronous sum (op 0) / max (op 1) of one host double over the ranks
 *   pgx_comm_fold_check  every rank must issue the same sequence of collectives with the same counts.  The first 8
 *                        tickets and every 16th after them (PGX_COMM_CHECK_FIRST / _EVERY) are preceded by a 32-byte
 *                        all-reduce(max) of {n, -n, h, -h}: n this collective's count, h a running hash of every count
 *                        so far and of every word folded in here (the caller's shared facts: window or block, rows,
 *                        switches).  Ranks that disagree get PGX_ERR_RUNTIME ("ranks out of step ...") from the
 *                        next call on the communicator instead of a hang or a corrupted sum; a check that does not
 *                        complete within PGX_COMM_CHECK_TIMEOUT_MS (60 s) fails the same way.
 *   pgx_comm_stats       tickets issued, agreement checks completed, current sequence hash (any pointer may be NULL)
 *   pgx_comm_quiesce     PGX_OK once everything handed to the communicator has completed, PGX_ERR_RUNTIME after
 *                        `timeout_ms` (a peer that died mid-collective never completes ours).  pgx_comm_destroy /
 *                        pgx_shutdown use it with PGX_COMM_EXIT_TIMEOUT_MS (10 s): past the deadline the communicator
 *                        is ABANDONED -- nothing of it is joined, synchronised or destroyed -- both return
 *                        PGX_ERR_RUNTIME and pgx_comm_abandoned() says 1: the process should exit non-zero. */
size_t pgx_comm_unique_id_bytes(void);
int pgx_comm_unique_id(void *id_host, size_t len);
int pgx_comm_init(int rank, int world, const void *id_host, size_t len);
int pgx_comm_info(int *rank, int *world);        /* world == 0: no communicator */
int pgx_comm_destroy(void);
int pgx_comm_quiesce(int timeout_ms);
int pgx_comm_abandoned(void);
int pgx_comm_fold_check(int64_t word);
int pgx_comm_stats(int64_t *issued, int64_t *checks, int64_t *hash);
int pgx_allreduce_sum(float *out, const float *in, size_t n, int64_t *ticket);
int pgx_allreduce_wait(int64_t ticket);
int pgx_allreduce_scalar_host(double *value_host, int op);

#ifdef __cplusplus
}
#endif
#endif /* PYGMU_HIP_H */
