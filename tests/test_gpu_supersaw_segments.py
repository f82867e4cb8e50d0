"""
GPU: a small bank of SuperSawPEs under a MixPE (a rank's share of the sharded 512-voice SuperSaw mix) rendered by the
fused bank kernels in concurrent TIME SEGMENTS (integrator levels from the closed form of the leaky integrator's
response to the BLIT harmonics, blit_saw_pe.py:196-235) against the same bank rendered oscillator by oscillator
(pgx_blitsaw + pgx_supersaw_sum, whose samples are the single PE's):
* pgx_supersaw_bank_seg (8 frames per thread, k_blitsaw's own phase sums replayed): the same float32 samples up to a
  rounding flip in ~1e-7 of them;
* pgx_supersaw_wide (16 frames per thread, phases frac(phase0 + (i+1) inc), voices not rounded to float32 before the
  sum): <= 1e-6 of the peak (measured ~1e-7: one or two float32 ulps).
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mix(pg, count, voices=7, base=55.0, channels=1):
    return pg.MixPE(*[pg.SuperSawPE(frequency=base * 2 ** (i / 12.0), voices=voices, detune_cents=20.0, seed=i,
                                    channels=channels) for i in range(count)])


def _run(segmented, count, blocks, wide=False, **kw):
    import pygmu2_amd as pg
    from pygmu2_amd import voice_bank
    pg.set_sample_rate(48000)
    keep = voice_bank.SEGMENTED_SUPERSAW, voice_bank.PREFETCH_SUPERSAW_VOICES, voice_bank.WIDE_SUPERSAW
    voice_bank.SEGMENTED_SUPERSAW = segmented
    voice_bank.WIDE_SUPERSAW = wide
    voice_bank.PREFETCH_SUPERSAW_VOICES = False        # (the pipelined path keeps its states one block ahead)
    try:
        mix = _mix(pg, count, **kw)
        r = pg.NullRenderer(sample_rate=48000)
        r.set_source(mix)
        r.start()
        bank = mix._voice_bank()
        assert bank and bank.root.segmented(blocks[0][1]) == segmented and bank.root.wide() == wide
        outs = [mix.render(s, n).data.copy() for s, n in blocks]
        state = bank.root.state.to_host().copy()
        r.stop()
        return outs, state
    finally:
        voice_bank.SEGMENTED_SUPERSAW, voice_bank.PREFETCH_SUPERSAW_VOICES, voice_bank.WIDE_SUPERSAW = keep


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("count,kw", [(64, {}), (16, dict(voices=3)), (100, dict(voices=5, base=27.5, channels=2))])
def test_time_segments_render_the_sequential_bank(count, kw, wide):
    blocks = [(0, 48_000), (48_000, 48_000), (96_000, 30_001), (500_000, 48_000)]       # a stream, then a seek (reset)
    got, st_got = _run(True, count, blocks, wide=wide, **kw)
    want, st_want = _run(False, count, blocks, **kw)
    for g, w in zip(got, want):
        peak = float(np.max(np.abs(w)))
        err = float(np.max(np.abs(g.astype(np.float64) - w)))
        assert err <= 1e-6 * peak, (err, peak)
        if not wide:
            assert np.mean(g != w) < 2e-3
    # 8 frames per thread: phases are the same additions, integrator levels agree to the closed form's ~1e-14;
    # 16 frames per thread: phases are products, not running sums (~1e-10 apart after 1e5 frames)
    assert np.allclose(st_got, st_want, rtol=0, atol=1e-9 if wide else 1e-11)


def test_full_bank_of_512_in_the_wide_form():
    """512 instances (one time segment each): pgx_supersaw_wide against pgx_supersaw_bank, whose samples are the
    per-voice ones bit for bit (test_gpu_voice_bank)."""
    import pygmu2_amd as pg
    from pygmu2_amd import voice_bank
    pg.set_sample_rate(48000)
    blocks = [(0, 48_000), (48_000, 20_000), (68_000, 4097)]

    def run(wide):
        keep = voice_bank.WIDE_SUPERSAW
        voice_bank.WIDE_SUPERSAW = wide
        try:
            mix = pg.MixPE(*[pg.SuperSawPE(frequency=55.0 * 2 ** (i / 96.0), voices=7, detune_cents=20.0, seed=i)
                             for i in range(512)])
            r = pg.NullRenderer(sample_rate=48000)
            r.set_source(mix)
            r.start()
            assert mix._voice_bank().root.wide() == wide
            outs = [mix.render(s, n).data.copy() for s, n in blocks]
            r.stop()
            return outs
        finally:
            voice_bank.WIDE_SUPERSAW = keep

    for g, w in zip(run(True), run(False)):
        peak = float(np.max(np.abs(w)))
        assert float(np.max(np.abs(g.astype(np.float64) - w))) <= 1e-6 * peak


def test_segment_plan_and_fallbacks():
    import pygmu2_amd as pg
    from pygmu2_amd import device
    lib = device.ensure_init()
    assert lib.pgx_supersaw_bank_segments(64, 48_000) == 4          # 12 tiles of 4096 frames, 3 per segment
    assert lib.pgx_supersaw_bank_segments(512, 48_000) == 1
    assert lib.pgx_supersaw_bank_segments(64, 4096) == 1
    assert lib.pgx_supersaw_wide_segments(64, 7, 48_000) == 4          # 12 tiles of 4096 frames too (4 waves x 16)
    assert lib.pgx_supersaw_wide_segments(512, 7, 48_000) == 1
    assert lib.pgx_supersaw_wide_segments(128, 7, 48_000) == 4          # two workgroups per CU beat one with six tiles
    pg.set_sample_rate(48000)
    # an explicit leak of 1.0 has no steady state: the bank keeps the oscillator-by-oscillator path
    mix = pg.MixPE(*[pg.SuperSawPE(frequency=110.0 + i, voices=3, seed=i) for i in range(8)])
    for pe in mix._inputs:
        for osc in pe._oscillators:
            osc._leak = 1.0
    bank = mix._voice_bank()
    assert bank and not bank.root.segmented(48_000)


@pytest.mark.parametrize("kind", ["blitsaw", "supersaw"])
def test_a_lone_oscillator_over_a_long_block(kind):
    """One BlitSawPE / SuperSawPE with scalar parameters rendered 300 000 frames at once (what a look-ahead window hands
    it): pgx_supersaw_wide with one instance in ~70 time segments -- against the same PE rendered in 4000-frame blocks
    (k_blitsaw: the reference's arithmetic) and against the oracle; the short block after it continues the stream."""
    import pygmu2_amd as pg
    from pygmu2_amd import blit_saw_pe
    from oracle import graph_eval
    from oracle.golden_cases import S
    pg.set_sample_rate(48000)
    if kind == "blitsaw":
        make = lambda: pg.BlitSawPE(frequency=173.3, amplitude=0.8)
        spec = S("BlitSawPE", frequency=173.3, amplitude=0.8)
    else:
        make = lambda: pg.SuperSawPE(frequency=98.0, voices=7, detune_cents=20.0, seed=3, channels=2)
        spec = S("SuperSawPE", frequency=98.0, voices=7, detune_cents=20.0, seed=3, channels=2)
    blocks = [(0, 300_000), (300_000, 4000), (304_000, 50_000)]

    def run(wide, cut):
        keep = blit_saw_pe.WIDE_LONG_RENDERS
        blit_saw_pe.WIDE_LONG_RENDERS = wide
        try:
            pe = make()
            r = pg.NullRenderer(sample_rate=48000)
            r.set_source(pe)
            r.start()
            outs = []
            for s, n in blocks:
                parts = [pe.render(p, min(cut, s + n - p)).data.copy() for p in range(s, s + n, cut)]
                outs.append(np.concatenate(parts))
            r.stop()
            return outs
        finally:
            blit_saw_pe.WIDE_LONG_RENDERS = keep

    got = run(True, 10 ** 9)
    want = run(False, 4000)
    g = graph_eval.Node(spec, 48000)
    for a, b, (s, n) in zip(got, want, blocks):
        peak = float(np.max(np.abs(b)))
        assert a.shape == b.shape
        assert float(np.max(np.abs(a.astype(np.float64) - b))) <= 1e-6 * peak
    ref = np.concatenate([g.render(p, 4000) for p in range(0, 300_000, 4000)])
    assert float(np.max(np.abs(got[0].astype(np.float64) - ref))) <= 1e-5 * float(np.max(np.abs(ref)))


def test_small_bank_windows_hand_out_the_block_by_block_samples(monkeypatch):
    """A bank of 40 SuperSawPEs streamed in equal blocks is rendered 2, 4, 8 blocks at a time and handed out as rows of
    the mixed window (voice_bank.BANK_WINDOWS); a seek inside a window, another block length and a restart put every
    node's state back.  Against the same bank block by block: <= 1e-6 of peak (the time segments of a longer render fall
    elsewhere; the closed-form carries agree to ~1e-14)."""
    import pygmu2_amd as pg
    from pygmu2_amd import voice_bank
    pg.set_sample_rate(48000)
    n = 12_288
    blocks = ([(i * n, n) for i in range(9)]
              + [(9 * n + 77, n), (10 * n + 77, n), (11 * n + 77, n)]
              + [(12 * n + 77, 5000), (12 * n + 5077, 5000), (12 * n + 10_077, 5000)]
              + [(0, n), (n, n), (2 * n, n)])

    def run(windows):
        monkeypatch.setattr(voice_bank, "BANK_WINDOWS", windows)
        mix = _mix(pg, 40, base=40.0)
        r = pg.NullRenderer(sample_rate=48000)
        r.set_source(mix)
        r.start()
        outs, opened = [], 0
        for s, m in blocks:
            outs.append(mix.render(s, m).data.copy())
            opened += mix._voice_bank().win is not None
        r.stop()
        return outs, opened

    got, opened = run(True)
    want, none = run(False)
    assert opened >= 8 and none == 0
    for a, b in zip(got, want):
        peak = float(np.max(np.abs(b)))
        assert a.shape == b.shape and float(np.max(np.abs(a.astype(np.float64) - b))) <= 1e-6 * peak


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_wide_bank_random_configurations(seed):
    """Random banks through pgx_supersaw_wide -- instance counts from 4 to 300 (one to many time segments, both sides of
    the fused-bank threshold), 1 to 16 oscillators per instance, frequencies from 1 Hz (M ~ 24 000 harmonics) to near
    Nyquist (M = 1), wide detune, mono and stereo, block lengths from 1 frame to 120 000 with odd tails, a seek -- against
    the oscillator-by-oscillator bank (k_blitsaw + ordered sum: the single PE's samples), <= 1e-6 of the peak."""
    import pygmu2_amd as pg
    from pygmu2_amd import voice_bank
    rng = np.random.default_rng(9000 + seed)
    pg.set_sample_rate(48000)
    count = int([4, 9, 33, 64, 130, 300][seed - 1])
    voices = int(rng.choice([1, 2, 3, 7, 11, 16]))
    channels = int(rng.choice([1, 2]))
    lo, hi = [(1.0, 40.0), (20.0, 2000.0), (2000.0, 16000.0)][int(rng.integers(0, 3))]
    freqs = np.exp(rng.uniform(np.log(lo), np.log(hi), count))
    detune = float(rng.choice([0.0, 5.0, 20.0, 50.0]))
    sizes = [int(rng.choice([1, 15, 17, 4095, 4096, 4097, 12_289, 48_000, 120_000])) for _ in range(5)]
    blocks, pos = [], 0
    for i, n in enumerate(sizes):
        if i == 3:
            pos += 777_777                               # a seek: the oscillators start over
        blocks.append((pos, n))
        pos += n

    def run(wide):
        keep = voice_bank.WIDE_SUPERSAW, voice_bank.SEGMENTED_SUPERSAW, voice_bank.BANK_WINDOWS
        voice_bank.WIDE_SUPERSAW = wide
        voice_bank.SEGMENTED_SUPERSAW = wide
        voice_bank.BANK_WINDOWS = False
        try:
            mix = pg.MixPE(*[pg.SuperSawPE(frequency=float(f), voices=voices, detune_cents=detune, seed=int(i),
                                           channels=channels) for i, f in enumerate(freqs)])
            r = pg.NullRenderer(sample_rate=48000)
            r.set_source(mix)
            r.start()
            root = mix._voice_bank().root
            used = root.wide()
            outs = [mix.render(s, n).data.copy() for s, n in blocks]
            r.stop()
            return outs, used
        finally:
            voice_bank.WIDE_SUPERSAW, voice_bank.SEGMENTED_SUPERSAW, voice_bank.BANK_WINDOWS = keep

    got, used = run(True)
    want, _ = run(False)
    if not used:
        pytest.skip("this bank keeps the oscillator-by-oscillator path (a harmonic count the rotation form excludes)")
    peak = max(float(np.max(np.abs(w))) for w in want)
    for (s, n), g, w in zip(blocks, got, want):
        assert g.shape == w.shape
        err = float(np.max(np.abs(g.astype(np.float64) - w)))
        assert err <= 1e-6 * peak, (s, n, err, peak, count, voices, channels, lo, hi, detune)
