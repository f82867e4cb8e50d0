"""GPU: read-ahead of pure sub-graphs (pygmu2_amd/read_ahead.py) hands out exactly the samples a
plain per-block render produces, for sequential, overlapping and random pulls."""

import numpy as np
import pytest

import pygmu2_amd as pg
from pygmu2_amd import read_ahead

pytestmark = pytest.mark.gpu


def _hello():
    mix = pg.MixPE(pg.SinePE(440.0), pg.SinePE(550.0), pg.SinePE(660.0))
    return pg.CropPE(pg.GainPE(mix, 0.3), 0, 8 * 44100)


def _pull(pe, requests):
    r = pg.NullRenderer(44100)
    r.set_source(pe)
    r.start()
    out = [pe.render(s, n).data.copy() for s, n in requests]
    r.stop()
    return out


def _loop():
    return pg.LoopPE(pg.CropPE(pg.SinePE(440.0), 0, 4410), crossfade_seconds=0.01)


def _window_max():
    return pg.WindowPE(pg.GainPE(pg.SinePE(3.0), pg.SinePE(441.0)), window=0.05)


@pytest.mark.parametrize("graph", ["c1", "hello", "loop", "window_max"])
def test_read_ahead_is_sample_identical(graph):
    pg.set_sample_rate(44100)
    make = {"c1": lambda: pg.GainPE(pg.SinePE(440.0, 1.0, 0.0, channels=2), gain=0.5), "hello": _hello,
            "loop": _loop, "window_max": _window_max}[graph]
    seq = [(i * 1024, 1024) for i in range(200)]
    seq += [(204800, 17), (204817, 1000), (100, 64), (164, 64), (228, 4096), (8 * 44100 - 100, 512)]
    read_ahead.set_enabled(True)
    try:
        a = _pull(make(), seq)
        pe = make()
        assert read_ahead.eligible(pe)
        read_ahead.set_enabled(False)
        b = _pull(make(), seq)
    finally:
        read_ahead.set_enabled(True)
    for x, y in zip(a, b):
        assert x.shape == y.shape and np.array_equal(x, y)


def test_read_ahead_skips_stateful_and_index_quirk_graphs():
    pg.set_sample_rate(44100)
    assert not read_ahead.eligible(pg.BiquadPE(pg.SinePE(440.0), 1000.0, 0.707))
    assert not read_ahead.eligible(pg.SinePE(frequency=pg.SinePE(5.0)))
    assert not read_ahead.eligible(pg.WindowPE(pg.SinePE(440.0), mode=pg.WindowMode.RMS))    # block-grouped sums
    assert not read_ahead.eligible(pg.LoopPE(pg.BiquadPE(pg.SinePE(440.0), 1000.0, 0.707), 0, 1000))
    # stateful chain streamed in small blocks still matches one big render within the budget
    a = pg.BiquadPE(pg.SinePE(440.0), 1000.0, 0.707)
    parts = _pull(a, [(i * 512, 512) for i in range(40)])
    whole = pg.BiquadPE(pg.SinePE(440.0), 1000.0, 0.707).render(0, 40 * 512).data
    assert np.max(np.abs(np.concatenate(parts) - whole)) <= 1e-6 * np.max(np.abs(whole))


def test_mix_with_bounded_inputs_is_not_read_ahead():
    """MixPE skips an input whose extent misses the REQUESTED window (mix_pe.py:81-85); with a hold-mode ArrayPE
    that makes the samples depend on the window, so such a MixPE must see exactly the caller's blocks."""
    import numpy as np
    import pygmu2_amd as pg
    from pygmu2_amd import read_ahead
    from oracle import graph_eval
    from oracle.golden_cases import S
    import spec_build
    spec = S("MixPE", inputs=[
        S("GainPE", source=S("ArrayPE", data={"rng": 688, "n": 2678, "ch": 1, "scale": 0.5}, extend_mode="hold_both"),
          gain=0.97),
        S("ArrayPE", data={"rng": 745, "n": 4314, "ch": 1, "scale": 0.5}, extend_mode="zero")])
    blocks = [[-361, 1], [-360, 257], [-103, 257], [154, 1024]]
    pg.set_sample_rate(48000)
    pe = spec_build.build(spec)
    assert not read_ahead.eligible(pe)
    unbounded = pg.MixPE(pg.SinePE(100.0), pg.GainPE(pg.SinePE(200.0), gain=0.5))
    assert read_ahead.eligible(unbounded)
    r = pg.NullRenderer(sample_rate=48000)
    r.set_source(pe)
    r.start()
    got = [pe.render(s, n).data for s, n in blocks]
    r.stop()
    g = graph_eval.Node(spec, 48000)
    for (s, n), out in zip(blocks, got):
        assert np.array_equal(out, g.render(s, n)), (s, n)


def test_identity_pe_streams_through_windows_below_and_beyond_2_24():
    """np.arange(start, start + n, dtype=float32) is filled as first + i * delta in float32: every index below 2^24 is a
    float32 whatever block it is rendered in, beyond that the samples depend on where a block begins.  A read-ahead window
    over IdentityPE is therefore cut for the caller's block length (round 4: pgx_ramp_blocks fills it block by block,
    each block the reference's own arange) and serves only pulls on that grid; anything else -- another length, a start
    off the grid -- is rendered for itself."""
    pg.set_sample_rate(44100)

    def want(start, n):
        return np.arange(start, start + n, dtype=np.float32).reshape(-1, 1)

    pe = pg.GainPE(pg.IdentityPE(), 2.0)
    assert read_ahead.eligible(pe)
    r = pg.NullRenderer(44100)
    r.set_source(pe)
    r.start()
    opened = False
    for i in range(40):
        got = pe.render(i * 1024, 1024).data
        opened = opened or pe.__dict__.get("_ra_win") is not None
        assert np.array_equal(got, 2.0 * want(i * 1024, 1024))
    assert opened
    base = (1 << 24) - 20 * 1024 - 7                       # a stream that walks across 2^24 (odd start: odd / even indices)
    for i in range(60):
        got = pe.render(base + i * 1024, 1024).data
        assert np.array_equal(got, 2.0 * want(base + i * 1024, 1024)), i
    win = pe.__dict__.get("_ra_win")
    assert read_ahead.eligible(pe) and win is not None and len(win) == 4 and win[3] == 1024     # still in windows
    # off the window's grid, or another length: rendered for itself, the reference's arange of THAT block
    at = base + 60 * 1024
    for start, n in ((at + 512, 1024), (at + 1024, 777), (at + 2048, 2048), (at + 4096, 1024), (at + 5120, 1024)):
        assert np.array_equal(pe.render(start, n).data, 2.0 * want(start, n)), (start, n)
    far = (1 << 30) + 12345                                 # far beyond: the fill steps by 128
    for i in range(30):
        assert np.array_equal(pe.render(far + i * 4096, 4096).data, 2.0 * want(far + i * 4096, 4096)), i
    r.stop()
