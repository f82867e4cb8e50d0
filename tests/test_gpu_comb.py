"""
GPU: CombPE (csrc/pgx_comb.hip) against the oracle's restatement of _comb_process_numba (comb_pe.py:26-113).

* scalar frequency, one time segment (<= 1024 steps per residue chain): the reference's loop operation for operation
  -> BIT-EXACT, whatever the block partition (the double-buffered ring carries the state);
* scalar frequency, long renders (concurrent time segments, carries through a float64 affine fold): <= 1e-7 of peak,
  and the ring left behind continues a stream exactly like the single-segment path does;
* frequency from a PE (time-parallel one-pole -> integer delays, ring in LDS; ring in HBM for a very low
  min_frequency): delays are integers, so the output is BIT-EXACT unless sr / f sits within ~1e-11 of a rounding
  tie;
* a bank of combs under a MixPE == the voices rendered one by one.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hip(spec, sr, blocks):
    import pygmu2_amd as pg
    import spec_build
    pg.set_sample_rate(sr)
    pe = spec_build.build(spec)
    r = pg.NullRenderer(sample_rate=sr)
    r.set_source(pe)
    r.start()
    outs = [pe.render(s, n).data.copy() for s, n in blocks]
    r.stop()
    return outs


def _oracle(spec, sr, blocks):
    from oracle import graph_eval
    g = graph_eval.Node(spec, sr)
    return [g.render(s, n) for s, n in blocks]


def _contig(sizes, start=0):
    out, pos = [], start
    for n in sizes:
        out.append((pos, n))
        pos += n
    return out


def _S():
    from oracle.golden_cases import S
    return S


NOISE = {"rng": 7, "n": 400_000, "ch": 2, "scale": 0.3}
NOISE1 = {"rng": 8, "n": 400_000, "ch": 1, "scale": 0.3}


@pytest.mark.parametrize("freq,fb,sr", [(440.0, 0.7, 44100), (30000.0, 0.99, 48000), (25.0, -0.9, 48000),
                                        (97.3, 0.5, 44100), (20.0, 0.995, 48000)])
def test_scalar_frequency_is_the_reference_loop(freq, fb, sr):
    S = _S()
    spec = S("CombPE", source=S("ArrayPE", data=NOISE), frequency=freq, feedback=fb)
    blocks = _contig([512, 17, 23, 4096, 1, 3000, 41, 20_000])
    got, want = _hip(spec, sr, blocks), _oracle(spec, sr, blocks)
    for i, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), (i, int(np.sum(g != w)), float(np.max(np.abs(g.astype(np.float64) - w))))


def test_scalar_frequency_feedback_stream():
    S = _S()
    spec = S("CombPE", source=S("ArrayPE", data=NOISE1), frequency=330.0,
             feedback=S("SinePE", frequency=1.3, amplitude=1.4))          # clamps at +-0.995 on the way
    blocks = _contig([3000, 19, 6000])
    got, want = _hip(spec, 48000, blocks), _oracle(spec, 48000, blocks)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


@pytest.mark.parametrize("freq,ch", [(440.0, 1), (220.0, 2), (12000.0, 2), (30.0, 1)])
def test_long_render_runs_in_time_segments(freq, ch):
    """400 000 frames in one call: 4 000 steps per chain at 440 Hz -> segmented.  Then a short block: the ring the
    segmented render left continues the stream."""
    S = _S()
    spec = S("CombPE", source=S("ArrayPE", data=NOISE if ch == 2 else NOISE1), frequency=freq, feedback=0.9)
    blocks = _contig([300_000, 5000, 90_000])
    got, want = _hip(spec, 44100, blocks), _oracle(spec, 44100, blocks)
    for g, w in zip(got, want):
        peak = float(np.max(np.abs(w)))
        err = float(np.max(np.abs(g.astype(np.float64) - w)))
        assert err <= 1e-7 * peak, (err, peak)
        assert np.mean(g != w) < 1e-3


def test_long_render_with_feedback_stream():
    S = _S()
    spec = S("CombPE", source=S("ArrayPE", data=NOISE1), frequency=1000.0,
             feedback=S("SinePE", frequency=0.7, amplitude=0.9))
    blocks = _contig([350_000, 2000])
    got, want = _hip(spec, 48000, blocks), _oracle(spec, 48000, blocks)
    for g, w in zip(got, want):
        peak = float(np.max(np.abs(w)))
        assert float(np.max(np.abs(g.astype(np.float64) - w))) <= 1e-7 * peak


def _sweep(lo, hi, hz):
    S = _S()
    return S("MixPE", inputs=[S("ConstantPE", value=0.5 * (lo + hi)), S("SinePE", frequency=hz, amplitude=0.5 * (hi - lo))])


@pytest.mark.parametrize("min_f,smooth", [(20.0, 2400), (30.0, 200), (2.0, 50), (20.0, 1)])
def test_frequency_stream_delays_are_exact(min_f, smooth):
    S = _S()
    spec = S("CombPE", source=S("ArrayPE", data=NOISE), frequency=_sweep(60.0, 900.0, 2.5), feedback=0.8,
             min_frequency=min_f, smoothing_samples=smooth)
    blocks = _contig([5000, 17, 44_100, 3, 9000])
    got, want = _hip(spec, 48000, blocks), _oracle(spec, 48000, blocks)
    for i, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), (i, int(np.sum(g != w)))


def test_frequency_and_feedback_streams_below_min_frequency():
    S = _S()
    spec = S("CombPE", source=S("ArrayPE", data=NOISE1), frequency=_sweep(-50.0, 400.0, 3.0),     # clamps at min_f
             feedback=S("SinePE", frequency=0.9, amplitude=1.2), min_frequency=25.0, smoothing_samples=480)
    blocks = _contig([20_000, 20_000])
    got, want = _hip(spec, 44100, blocks), _oracle(spec, 44100, blocks)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_bank_of_combs_matches_voices_one_by_one():
    import pygmu2_amd as pg
    from pygmu2_amd import voice_bank
    pg.set_sample_rate(48000)

    def voices():
        return [pg.CombPE(pg.BlitSawPE(frequency=55.0 * 2 ** (i / 7.0)), frequency=40.0 * 2 ** (i / 5.0),
                          feedback=0.3 + 0.02 * i) for i in range(24)]

    from pygmu2_amd import blit_saw_pe

    def run(banked):
        keep = voice_bank.MIN_VOICES, voice_bank.SEGMENTED_SUPERSAW, blit_saw_pe.WIDE_LONG_RENDERS
        voice_bank.MIN_VOICES = 4 if banked else 10 ** 9
        voice_bank.SEGMENTED_SUPERSAW = False      # the oscillators in front of the combs: not what is compared here
        blit_saw_pe.WIDE_LONG_RENDERS = False      # (... bank or lone: k_blitsaw's samples either way)
        try:
            mix = pg.MixPE(*voices())
            r = pg.NullRenderer(sample_rate=48000)
            r.set_source(mix)
            r.start()
            assert bool(mix._voice_bank()) == banked
            outs = [mix.render(s, n).data.copy() for s, n in _contig([4800, 100_000, 777])]
            r.stop()
            return outs
        finally:
            voice_bank.MIN_VOICES, voice_bank.SEGMENTED_SUPERSAW, blit_saw_pe.WIDE_LONG_RENDERS = keep

    for a, b in zip(run(True), run(False)):
        assert np.array_equal(a, b)


def test_look_ahead_window_of_a_comb_stream():
    """44 100-frame blocks through look-ahead windows (a window is one long, time-segmented render) against the
    oracle rendering block by block."""
    S = _S()
    spec = S("CombPE", source=S("SinePE", frequency=330.0), frequency=440.0, feedback=0.7)
    blocks = _contig([44_100] * 12)
    got, want = _hip(spec, 44100, blocks), _oracle(spec, 44100, blocks)
    for g, w in zip(got, want):
        peak = float(np.max(np.abs(w)))
        assert float(np.max(np.abs(g.astype(np.float64) - w))) <= 1e-6 * peak


def test_frequency_stream_over_a_long_window():
    """300 000 frames in one call with a PE frequency: the control one-pole runs in several 65 536-sample segments
    (two launches), the ring walks the whole block.  Delays are integers: bit-exact against the oracle."""
    S = _S()
    spec = S("CombPE", source=S("ArrayPE", data=NOISE1), frequency=_sweep(80.0, 700.0, 1.7), feedback=0.6,
             smoothing_samples=960)
    blocks = _contig([300_000, 4000, 70_000])
    got, want = _hip(spec, 48000, blocks), _oracle(spec, 48000, blocks)
    for i, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), (i, int(np.sum(g != w)))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PGX_FUZZ_COMB", "16"))))
def test_random_combs_against_the_oracle(seed):
    """Random CombPEs -- scalar or PE-driven frequency (swept through and below min_frequency), scalar or PE-driven
    feedback, smoothing from 1 to 4800 samples, min_frequency from 2 Hz (ring in HBM) to 60 Hz, mono / stereo noise or an
    oscillator in front, and pulls from 1 frame to 300 000 (one segment, many segments, streams of equal small blocks
    through look-ahead windows, a seek) -- every block against the oracle's loop (bit-exact where the docstring above
    says so; this test does not tell the cases apart)."""
    S = _S()
    rng = np.random.default_rng(77_000 + seed)
    sr = int(rng.choice([44100, 48000]))
    ch = int(rng.choice([1, 2]))
    src = [S("ArrayPE", data=NOISE1 if ch == 1 else NOISE),
           S("BlitSawPE", frequency=float(rng.uniform(50.0, 900.0)), channels=ch),
           S("SinePE", frequency=float(rng.uniform(50.0, 5000.0)), amplitude=0.5, channels=ch)][int(rng.integers(0, 3))]
    kw = {}
    if rng.random() < 0.5:
        kw["frequency"] = float(np.exp(rng.uniform(np.log(25.0), np.log(20000.0))))
    else:
        lo, hi = sorted(float(v) for v in np.exp(rng.uniform(np.log(10.0), np.log(3000.0), 2)))
        kw["frequency"] = _sweep(lo if rng.random() < 0.8 else -lo, hi, float(rng.uniform(0.2, 8.0)))
        kw["min_frequency"] = float(rng.choice([2.0, 20.0, 30.0, 60.0]))
        kw["smoothing_samples"] = int(rng.choice([1, 50, 480, 2400, 4800]))
    kw["feedback"] = (float(rng.uniform(-0.98, 0.98)) if rng.random() < 0.6
                      else S("SinePE", frequency=float(rng.uniform(0.1, 5.0)), amplitude=float(rng.uniform(0.3, 1.2))))
    spec = S("CombPE", source=src, **kw)
    pattern = int(rng.integers(0, 3))
    if pattern == 0:
        sizes = [int(rng.choice([1, 17, 1000, 5000, 44_100])) for _ in range(5)]
    elif pattern == 1:
        sizes = [int(rng.choice([44_100, 100_001, 300_000])), 2000, int(rng.choice([1, 48_000]))]
    else:
        sizes = [int(rng.choice([256, 1024, 4096]))] * int(rng.integers(12, 40))
    blocks = _contig(sizes, start=int(rng.choice([0, 0, 777])))
    if rng.random() < 0.4 and len(blocks) > 3:                       # a seek in the middle of the stream
        k = len(blocks) // 2
        blocks = blocks[:k] + [(s + 100_000, n) for s, n in blocks[k:]]
    got, want = _hip(spec, sr, blocks), _oracle(spec, sr, blocks)
    peak = max(float(np.max(np.abs(w))) for w in want) or 1.0
    # the comb itself: 1e-6 (float32 noise in, exact or 1e-7 out).  Behind an oscillator it repeats what the oscillator's
    # long-block kernels are off by (<= 1e-6 of THEIR peak) with its own gain, 1 / (1 - |feedback|): the 1e-5 budget
    tol = 1e-6 if src["pe"] == "ArrayPE" else 1e-5
    for i, ((s, n), g, w) in enumerate(zip(blocks, got, want)):
        err = float(np.max(np.abs(g.astype(np.float64) - w)))
        assert err <= tol * peak, (i, s, n, err, peak, sr, ch, kw)


# ---------------------------------------------------------------------------- rounding ties in the delays (index work)
def _literal_levels(raw, alpha, min_f):
    """comb_pe.py:61-68 in Python floats: the smoothed frequency after every sample."""
    out = np.empty(len(raw))
    sm = -1.0
    for i, v in enumerate(raw):
        v = float(v)
        v = v if v >= min_f else min_f
        sm = v if sm < 0.0 else sm + (v - sm) * alpha
        out[i] = sm
    return out


def _ulps(raw, lo, hi, k):
    """raw[lo:hi] moved by k float32 ulps (positive floats: the integer representation is monotonic)."""
    raw = raw.copy()
    bits = raw.view(np.int32)
    bits[lo:hi] += k
    return raw


def plant_a_tie(sr=44100, n=30_000, smoothing=2400, min_f=20.0, tie=220.5):
    """A float32 frequency stream (a slow ramp through sr / tie) whose smoothed level puts sr / f within ~2e-12 of the
    rounding tie at one sample: coarse to fine -- a run of samples moved by whole ulps, a shorter run by one ulp, then
    one ulp up at one sample and one ulp down at another (steps of ~3e-12 in sr / f)."""
    alpha = 1.0 / smoothing
    f_tie = sr / tie
    raw = np.linspace(f_tie - 10.0, f_tie + 10.0, n).astype(np.float32)
    q = sr / _literal_levels(raw, alpha, min_f)
    t = int(np.argmax(q < tie))                                   # the first sample below the tie
    assert 6000 < t < n - 10
    ulp = float(np.spacing(np.float32(f_tie)))

    def residual(r):
        sm = _literal_levels(r[:t + 1], alpha, min_f)[t]
        return sr / sm - tie, sm

    for _ in range(40):
        r, sm = residual(raw)
        if abs(r) < 2e-12:
            break
        need = r * sm / (sr / sm)                                 # change of the level that cancels r (dq/dsm = -q/sm)
        if abs(need) > 0.8 * ulp:
            raw = _ulps(raw, t - 4800, t + 1, int(round(need / (ulp * (1.0 - (1.0 - alpha) ** 4801)))))
        elif abs(need) > alpha * ulp:
            length = int(round(np.log(1.0 - abs(need) / ulp) / np.log(1.0 - alpha)))
            raw = _ulps(raw, t + 1 - max(length, 1), t + 1, 1 if need > 0 else -1)
        else:
            # +1 ulp at lag a, -1 ulp at lag b: alpha ulp ((1 - alpha)^a - (1 - alpha)^b)
            a = int(np.random.default_rng(_).integers(0, 40))
            target = (1.0 - alpha) ** a - abs(need) / (alpha * ulp)
            if target <= 0.05:
                raw = _ulps(raw, t, t + 1, 1 if need > 0 else -1)
                continue
            b = int(round(np.log(target) / np.log(1.0 - alpha)))
            if b <= a:
                b = a + 1
            s = 1 if need > 0 else -1
            raw = _ulps(_ulps(raw, t - a, t - a + 1, s), t - b, t - b + 1, -s)
    r, _sm = residual(raw)
    return raw, t, r


def test_rounding_ties_in_the_delays_take_the_literal_chain():
    """delays are index work.  A stream planted so that sr / f sits ~1e-12 from a rounding tie at one
    sample (closer than the time-parallel level and the reference's chain agree): k_comb_delays flags the block, the
    literal one-lane chain makes its delays again, and the output is the reference's bit for bit."""
    import pygmu2_amd as pg
    from oracle import pe_oracle as O
    sr, n = 44100, 30_000
    raw, t, r = plant_a_tie(sr, n)
    assert abs(r) < 1e-10, r                                      # (2e-12 as a rule; anything below 1e-9 raises the flag)
    rng = np.random.default_rng(5)
    x = (0.3 * rng.standard_normal((n, 1))).astype(np.float32)
    pg.set_sample_rate(sr)
    pe = pg.CombPE(pg.ArrayPE(x), frequency=pg.ArrayPE(raw.reshape(-1, 1)), feedback=0.7, min_frequency=20.0,
                   smoothing_samples=2400)
    rr = pg.NullRenderer(sample_rate=sr)
    rr.set_source(pe)
    rr.start()
    blocks = [(0, t - 3000), (t - 3000, 9000), (t + 6000, n - t - 6000)]           # the tie sits in the second block
    got = [pe.render(s, m).data.copy() for s, m in blocks]
    from pygmu2_amd import look_ahead
    look_ahead.settle(pe)                         # (a window opened at the second pull ran on past the stream's end)
    state = pe._state.to_host()
    rr.stop()
    st = O.comb_state(1, sr, 20.0)
    want = [O.comb(st, x[s:s + m], raw[s:s + m].astype(np.float64), 0.7, 20.0, 2400, sr) for s, m in blocks]
    for i, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), (i, int(np.sum(g != w)))
    assert int(state[3:4].view(np.int64)[0]) == 1, "exactly the block with the planted tie is redone"
    assert state[0] == st["sm"], "the carried level is the literal chain's, bit for bit"
    assert int(state[2:3].view(np.int32)[0]) == 0, "the flag is lowered again"
