"""
GPU: BiquadPE(SinePE) as ONE launch (pgx_biquad_sine: the sine is generated in registers inside the settled filter
kernel, csrc/pgx_scan.hip k_biquad_settled<.., SINE>) against the two-PE path and against the oracle
(np.sin + scipy.signal.lfilter, biquad_pe.py:383-404 over sine_pe.py:159-175).
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _chain(pg, freq=440.0, amp=1.0, phase=0.0, cutoff=1000.0, q=0.707, mode="lowpass"):
    return pg.BiquadPE(pg.SinePE(frequency=freq, amplitude=amp, phase=phase), frequency=cutoff, q=q,
                       mode=pg.BiquadMode(mode))


def _render(fuse, blocks, sr=44100, **kw):
    import pygmu2_amd as pg
    from pygmu2_amd import biquad_pe, look_ahead
    pg.set_sample_rate(sr)
    keep = biquad_pe.FUSE_SINE_SOURCE
    biquad_pe.FUSE_SINE_SOURCE = fuse
    look_ahead.set_enabled(False)
    try:
        pe = _chain(pg, **kw)
        r = pg.NullRenderer(sample_rate=sr)
        r.set_source(pe)
        r.start()
        outs = [pe.render(s, n).data.copy() for s, n in blocks]
        r.stop()
        return outs
    finally:
        biquad_pe.FUSE_SINE_SOURCE = keep
        look_ahead.set_enabled(True)


@pytest.mark.parametrize("kw", [dict(), dict(freq=3000.3, amp=0.3, phase=1.1, cutoff=2500.0, q=2.0, mode="bandpass"),
                                dict(freq=5500.0, cutoff=3000.0, mode="highpass"),
                                dict(freq=700.0, cutoff=1000.0, q=1.0, mode="peaking")])
def test_fused_chain_equals_the_two_launch_chain(kw):
    blocks = [(0, 1_000_000), (1_000_000, 300_001), (1_300_001, 50_000), (10 ** 9, 2_000_000)]
    fused, plain = _render(True, blocks, **kw), _render(False, blocks, **kw)
    for a, b in zip(fused, plain):
        peak = float(np.max(np.abs(b)))
        err = float(np.max(np.abs(a.astype(np.float64) - b)))
        assert err <= 5e-7 * peak, (err, peak)          # a float32 sine sample rounding the other way now and then


def test_fused_chain_against_the_oracle():
    from oracle import pe_oracle as O
    n = 1_000_000
    got = _render(True, [(0, n), (n, n)])
    st = O.biquad_state(1)
    for k in range(2):
        want = O.biquad_const(st, O.sine_pure(k * n, n, 440.0, sr=44100), 1000.0, 0.707, "lowpass", 0.0, 44100)
        peak = float(np.max(np.abs(want)))
        assert float(np.max(np.abs(got[k].astype(np.float64) - want))) <= 1e-6 * peak


def test_filters_that_pass_rounding_noise_rather_than_the_tone_keep_two_launches():
    """A high-pass three octaves above the tone answers mostly to the float32 rounding of its input: there the
    sine has to be the separate SinePE's float32 samples, and the chain is the two-launch chain bit for bit."""
    import pygmu2_amd as pg
    kw = dict(freq=55.0, cutoff=8000.0, mode="highpass")
    pg.set_sample_rate(44100)
    assert _chain(pg, **kw)._render_sine_source(0, 1_000_000) is None
    blocks = [(0, 1_000_000), (10 ** 9, 1_000_000)]
    for a, b in zip(_render(True, blocks, **kw), _render(False, blocks, **kw)):
        assert np.array_equal(a, b)


def test_fused_path_is_taken_and_declined():
    import pygmu2_amd as pg
    from pygmu2_amd import device
    lib = device.ensure_init()
    pg.set_sample_rate(44100)
    pe = _chain(pg)
    assert pe._render_sine_source(0, 1_000_000) is not None
    assert pe._render_sine_source(0, 1024) is not None                  # a short block: one workgroup, carried state
    slow = pg.BiquadPE(pg.SinePE(frequency=440.0), frequency=20.0, q=10.0)   # forgets too slowly: the exact pair
    assert slow._render_sine_source(0, 1_000_000) is None
    assert pe._render_sine_source(10 ** 13, 1_000_000) is None          # phase beyond the bounded sine's range
    stereo = pg.BiquadPE(pg.SinePE(frequency=440.0, channels=2), frequency=1000.0, q=0.707)
    assert stereo._render_sine_source(0, 1_000_000) is None
    fm = pg.BiquadPE(pg.SinePE(frequency=pg.SinePE(frequency=2.0, amplitude=100.0)), frequency=1000.0, q=0.707)
    assert fm._render_sine_source(0, 1_000_000) is None
    assert lib.pgx_biquad_sine_supported(1_000_000, 1024) == 1 and lib.pgx_biquad_sine_supported(1_000_000, 0) == 0


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PGX_FUZZ_C2", "16"))))
def test_random_chains_against_the_oracle(seed):
    """BiquadPE(SinePE) with random tone, filter and block lengths -- short blocks (one workgroup, carried state), blocks
    of millions of frames (1024 workgroups, anchors turned from tile to tile), a seek to a far start -- whichever form the
    PE picks (fused, or the two-launch chain where the fused form declines), every block against np.sin + lfilter."""
    from oracle import pe_oracle as O
    import pygmu2_amd as pg
    from pygmu2_amd import look_ahead
    rng = np.random.default_rng(12_000 + seed)
    sr = int(rng.choice([22050, 44100, 48000, 96000]))
    freq = float(np.exp(rng.uniform(np.log(20.0), np.log(0.45 * sr))))
    amp, phase = float(rng.uniform(0.05, 1.0)), float(rng.uniform(-3.0, 3.0))
    mode = str(rng.choice(["lowpass", "highpass", "bandpass", "notch", "allpass", "peaking", "lowshelf", "highshelf"]))
    cutoff = float(np.exp(rng.uniform(np.log(30.0), np.log(0.45 * sr))))
    q, gain_db = float(np.exp(rng.uniform(np.log(0.3), np.log(8.0)))), float(rng.uniform(-12.0, 12.0))
    sizes = [int(rng.choice([1, 1000, 4097, 65_536, 1_000_000, 3_300_001])) for _ in range(3)]
    blocks, pos = [], int(rng.choice([0, 0, 12_345, 10 ** 9]))
    for i, n in enumerate(sizes):
        if i == 2 and rng.random() < 0.5:
            pos += 10 ** 8                                       # a seek: the filter state carries on, as in the reference
        blocks.append((pos, n))
        pos += n
    pg.set_sample_rate(sr)
    look_ahead.set_enabled(False)
    try:
        pe = pg.BiquadPE(pg.SinePE(frequency=freq, amplitude=amp, phase=phase), frequency=cutoff, q=q,
                         mode=pg.BiquadMode(mode), gain_db=gain_db)
        r = pg.NullRenderer(sample_rate=sr)
        r.set_source(pe)
        r.start()
        got = [pe.render(s, n).data.copy() for s, n in blocks]
        r.stop()
    finally:
        look_ahead.set_enabled(True)
    st = O.biquad_state(1)
    peak = 0.0
    wants = []
    for s, n in blocks:
        wants.append(O.biquad_const(st, O.sine_pure(s, n, freq, amp, phase, sr=sr), cutoff, q, mode, gain_db, sr))
        peak = max(peak, float(np.max(np.abs(wants[-1]))))
    for (s, n), g, w in zip(blocks, got, wants):
        err = float(np.max(np.abs(g.astype(np.float64) - w)))
        assert err <= 1e-6 * peak + 1e-9, (s, n, err, peak, sr, freq, amp, phase, mode, cutoff, q, gain_db)
