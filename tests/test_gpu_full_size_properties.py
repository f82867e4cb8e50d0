"""GPU: BASELINE.json's sharded configurations at their FULL voice counts and block size, through the property the
sharding rests on: the mix is a plain sum (mix_pe.py:91-94), so the shares rendered by the G ranks (inputs
i = r mod G, each with its own state) add up to the unsharded mix.  One GPU plays every rank in turn; this also
runs the kernel variants a rank's smaller share selects (several workgroups per oscillator below 128 voices, a
workgroup per envelope in the ADSR walk) against the bank-wide ones at full size.  Two consecutive 48 000-frame
blocks, so carried state is part of it."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BLOCK = 48_000


def _render(pg, voices, blocks=2):
    root = pg.MixPE(*voices) if len(voices) > 1 else voices[0]
    r = pg.NullRenderer(sample_rate=48000)
    r.set_source(root)
    r.start()
    out = [root.render(i * BLOCK, BLOCK).data.astype(np.float64) for i in range(blocks)]
    r.stop()
    return np.concatenate(out)


@pytest.mark.parametrize("config,n_voices,world", [("c5", 512, 8), ("c4", 64, 4), ("supersaw", 512, 8)])
def test_rank_shares_add_up_to_the_full_mix(config, n_voices, world):
    import pygmu2_amd as pg
    from pygmu2_amd.sharding import c4_voice, c5_voice, shard_indices, supersaw_voice
    pg.set_sample_rate(48000)
    make = {"c5": c5_voice, "c4": c4_voice, "supersaw": supersaw_voice}[config]
    full = _render(pg, [make(pg, i) for i in range(n_voices)])
    total = np.zeros_like(full)
    for rank in range(world):
        total += _render(pg, [make(pg, i) for i in shard_indices(n_voices, rank, world)])
    peak = float(np.max(np.abs(full)))
    assert peak > 0.1 and np.all(np.isfinite(full))
    # float32 sums in a different grouping: n_voices terms of magnitude <= 1
    assert float(np.max(np.abs(total - full))) <= 1e-5 * peak, (config, float(np.max(np.abs(total - full))), peak)


def test_c3_full_length_convolution_properties():
    """BASELINE config 3 at its full length (1 440 000 stereo frames x 65 536 taps, fft_size 131072): the result
    does not depend on the block partition (whole vs 65 537-frame blocks), halving the input
    halves the output exactly (a power-of-two scale passes through the float64 transforms untouched), and a late
    window agrees with numpy's float64 convolution of just the samples that reach it."""
    import pygmu2_amd as pg
    from scipy.signal import fftconvolve
    pg.set_sample_rate(48000)
    T, L = 1_440_000, 65_536
    x = (np.random.default_rng(0).standard_normal((T, 2)) * 0.1).astype(np.float32)
    h = (np.random.default_rng(1).standard_normal(L) * np.exp(-np.arange(L) / 8000.0)).astype(np.float32)

    def run(src, sizes):
        pe = pg.ConvolvePE(pg.ArrayPE(src), pg.ArrayPE(h), fft_size=131072)
        r = pg.NullRenderer(sample_rate=48000)
        r.set_source(pe)
        r.start()
        pos, parts = 0, []
        for n in sizes:
            parts.append(pe.render(pos, n).data)
            pos += n
        r.stop()
        return np.concatenate(parts)

    whole = run(x, [T])
    blocks = [65_537] * (T // 65_537)
    blocks.append(T - sum(blocks))
    chunked = run(x, blocks)
    peak = float(np.max(np.abs(whole)))
    assert peak > 1.0 and np.all(np.isfinite(whole))
    assert float(np.max(np.abs(whole.astype(np.float64) - chunked))) <= 1e-6 * peak
    assert np.array_equal(run(x * np.float32(0.5), [T]), whole * np.float32(0.5))
    w0, w1 = T - 4000, T
    for c in range(2):
        seg = x[w0 - (L - 1):w1, c].astype(np.float64)
        want = fftconvolve(seg, h.astype(np.float64))[L - 1:L - 1 + (w1 - w0)]
        assert float(np.max(np.abs(whole[w0:w1, c] - want))) <= 1e-5 * peak


def test_c2_full_length_biquad_properties():
    """BASELINE config 2 at its full length (BiquadPE low-pass on SinePE, one render of 1 000 000 frames): the same
    samples come out of 1024-frame blocks, of uneven blocks and of one call; halving the source amplitude halves the
    output exactly; and a late window agrees with scipy's lfilter started 100 000 frames earlier from rest (the
    section has long forgotten where it started)."""
    import pygmu2_amd as pg
    from scipy.signal import lfilter
    from oracle import pe_oracle as O
    pg.set_sample_rate(44100)
    n = 1_000_000

    def run(sizes, amplitude=1.0):
        pe = pg.BiquadPE(pg.SinePE(frequency=440.0, amplitude=amplitude), frequency=1000.0, q=0.707,
                         mode=pg.BiquadMode.LOWPASS)
        r = pg.NullRenderer(sample_rate=44100)
        r.set_source(pe)
        r.start()
        pos, parts = 0, []
        for k in sizes:
            parts.append(pe.render(pos, k).data)
            pos += k
        r.stop()
        return np.concatenate(parts)[:, 0]

    whole = run([n])
    peak = float(np.max(np.abs(whole)))
    small = [1024] * (n // 1024) + [n % 1024]
    uneven = [333_333, 1, 65_537, 601_129]
    assert sum(uneven) == n
    for sizes in (small, uneven):
        assert float(np.max(np.abs(run(sizes).astype(np.float64) - whole))) <= 1e-6 * peak, len(sizes)
    assert np.array_equal(run([n], amplitude=0.5), whole * np.float32(0.5))
    b0, b1, b2, a1, a2 = O.biquad_coeffs(np.array([1000.0]), np.array([0.707]), "lowpass", 0.0, 44100)
    b, a = np.array([b0[0], b1[0], b2[0]]), np.array([1.0, a1[0], a2[0]])
    w0 = 900_000
    x = O.sine_pure(w0 - 100_000, 200_000, 440.0, 1.0, 0.0, 44100, 1)[:, 0].astype(np.float64)
    want = lfilter(b, a, x)[100_000:]
    assert float(np.max(np.abs(whole[w0:] - want))) <= 1e-5 * peak


def test_c1_full_length_stream_matches_numpy():
    """BASELINE config 1 at its full length: 10 s of GainPE(SinePE stereo) in 1024-frame blocks (431 of them, the
    last one 672 frames) against the oracle evaluated in one piece."""
    import pygmu2_amd as pg
    from oracle import pe_oracle as O
    pg.set_sample_rate(44100)
    total = 441_000
    pe = pg.GainPE(pg.SinePE(frequency=440.0, amplitude=1.0, phase=0.0, channels=2), gain=0.5)
    r = pg.NullRenderer(sample_rate=44100)
    r.set_source(pe)
    r.start()
    pos, parts = 0, []
    while pos < total:
        k = min(1024, total - pos)
        parts.append(pe.render(pos, k).data)
        pos += k
    r.stop()
    got = np.concatenate(parts)
    want = O.gain_const(O.sine_pure(0, total, 440.0, 1.0, 0.0, 44100, 2), 0.5)
    assert got.shape == want.shape == (total, 2)
    assert float(np.max(np.abs(got.astype(np.float64) - want))) <= 1e-5 * 0.5


@pytest.mark.parametrize("config,n_voices,seconds", [("c5", 512, 60), ("c4", 64, 30)])
def test_sharded_configs_full_duration_restart_reproduces(config, n_voices, seconds):
    """BASELINE configs 4 and 5 over their full durations (30 s / 60 s in 48 000-frame blocks): finite, bounded
    by the voice count, and a stop / start renders the same samples again (every state returns to its origin)."""
    import pygmu2_amd as pg
    from pygmu2_amd.sharding import c4_voice, c5_voice
    pg.set_sample_rate(48000)
    make = c5_voice if config == "c5" else c4_voice
    root = pg.MixPE(*[make(pg, i) for i in range(n_voices)])
    r = pg.NullRenderer(sample_rate=48000)
    r.set_source(root)

    def run():
        r.start()
        checks = []
        for i in range(seconds):
            block = root.render(i * BLOCK, BLOCK).data
            assert np.all(np.isfinite(block)) and float(np.max(np.abs(block))) < n_voices
            checks.append((float(block.astype(np.float64).sum()), float(np.abs(block).max())))
            if i in (0, seconds - 1):
                checks.append(block.copy())
        r.stop()
        return checks

    first, second = run(), run()
    for a, b in zip(first, second):
        if isinstance(a, tuple):
            assert a == b
        else:
            assert np.array_equal(a, b)
