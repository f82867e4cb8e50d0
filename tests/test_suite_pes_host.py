"""CPU: constructor / property / extent / error behaviour of LoopPE, WindowPE, DynamicsPE and the compressor family,
as the reference's own tests pin it (test_loop_pe.py:20-126,360-378; test_window_pe.py:22-95;
test_dynamics_pe.py:24-124; test_compressor_pe.py:24-112,232-350).  Host logic only: nothing here renders."""

import numpy as np
import pytest

import pygmu2_amd as pg
from pygmu2_amd import (CompressorPE, ConstantPE, CropPE, DetectionMode, DynamicsMode, DynamicsPE, EnvelopePE,
                        ExpanderPE, LimiterPE, LoopPE, NullRenderer, PiecewisePE, SinePE, WindowMode, WindowPE)


@pytest.fixture(autouse=True)
def _rate():
    pg.set_sample_rate(44100)


def _ramp():
    return PiecewisePE([(0, 0.0), (100, 1.0)])


def test_loop_properties_and_repr():
    src = _ramp()
    loop = LoopPE(src)
    assert loop.source is src and loop.loop_start is None and loop.loop_end is None and loop.count is None
    assert loop.crossfade_seconds == 0.0 and loop.inputs() == [src] and loop.is_pure() is True
    loop = LoopPE(src, loop_start=10, loop_end=50, count=4, crossfade_seconds=0.01)
    assert (loop.loop_start, loop.loop_end, loop.count, loop.crossfade_seconds) == (10, 50, 4, 0.01)
    assert loop.crossfade_samples == min(int(round(0.01 * 44100)), 40 // 2)
    text = repr(LoopPE(src, loop_start=10, loop_end=50, count=4))
    assert all(part in text for part in ("LoopPE", "PiecewisePE", "10", "50", "count=4"))
    assert LoopPE(ConstantPE(1.0, channels=2), loop_end=100).channel_count() == 2


def test_loop_extent_and_errors():
    r = NullRenderer(sample_rate=44100)
    loop = LoopPE(_ramp())
    r.set_source(loop)
    assert (loop.extent().start, loop.extent().end) == (0, None)
    assert LoopPE(_ramp(), count=3).extent().end == 300
    assert LoopPE(_ramp(), loop_start=20, loop_end=60, count=5).extent().end == 200
    with pytest.raises(ValueError):
        LoopPE(_ramp(), crossfade_seconds=-0.1)
    with pytest.raises(ValueError, match="positive"):
        LoopPE(_ramp(), loop_start=50, loop_end=50)
    with pytest.raises(ValueError, match="infinite"):
        LoopPE(ConstantPE(1.0))
    LoopPE(ConstantPE(1.0), loop_start=0, loop_end=100)          # explicit end: fine


def test_window_properties():
    src = SinePE(frequency=440.0)
    w = WindowPE(src)
    assert w.source is src and w.window == 0.05 and w.mode == WindowMode.MAX and w.rectify is True
    w = WindowPE(src, window=0.02, mode=WindowMode.RMS, rectify=False)
    assert (w.window, w.mode, w.rectify) == (0.02, WindowMode.RMS, False)
    assert WindowPE(src, window=-1.0).window == 0.0              # clamped (test_window_pe.py:47-52)
    assert w.inputs() == [src] and w.is_pure() is True
    assert WindowPE(ConstantPE(1.0, channels=2)).channel_count() == 2
    cropped = CropPE(src, 100, 400)
    assert WindowPE(cropped).extent() == cropped.extent()
    text = repr(WindowPE(src, window=0.03, mode=WindowMode.MEAN))
    assert all(part in text for part in ("WindowPE", "SinePE", "0.03", "mean"))


def test_dynamics_properties_extent_and_makeup():
    src, env = ConstantPE(1.0), ConstantPE(0.5)
    d = DynamicsPE(src, env)
    assert (d.threshold, d.ratio, d.knee, d.mode, d.stereo_link) == (-20.0, 4.0, 0.0, DynamicsMode.COMPRESS, True)
    d = DynamicsPE(src, env, threshold=-10.0, ratio=8.0, knee=6.0, makeup_gain=3.0, mode=DynamicsMode.LIMIT,
                   stereo_link=False)
    assert (d.threshold, d.ratio, d.knee, d.makeup_gain, d.mode, d.stereo_link) == (-10.0, 8.0, 6.0, 3.0,
                                                                                    DynamicsMode.LIMIT, False)
    sine = SinePE(frequency=440.0)
    follower = EnvelopePE(sine)
    d = DynamicsPE(sine, follower)
    assert len(d.inputs()) == 2 and sine in d.inputs() and follower in d.inputs() and d.is_pure() is True
    assert DynamicsPE(ConstantPE(1.0, channels=2), env).channel_count() == 2
    auto = DynamicsPE(src, env, threshold=-20, ratio=4, makeup_gain="auto")
    assert auto.makeup_gain == pytest.approx(0.7 * 9.0)          # 12 dB over at 4:1 loses 9 dB; 70 % is given back
    assert DynamicsPE(src, env, mode=DynamicsMode.GATE).makeup_gain == 0.0
    text = repr(DynamicsPE(src, env, threshold=-20, ratio=4))
    assert all(part in text for part in ("DynamicsPE", "threshold=-20", "ratio=4"))
    disjoint = DynamicsPE(CropPE(ConstantPE(1.0), 0, 10), CropPE(ConstantPE(0.5), 20, 10))
    assert disjoint.extent().is_empty()


def test_dynamics_static_curve_matches_oracle_curve():
    """The host's scalar curve (automatic make-up gain) against the oracle's restatement of _compute_gain_db."""
    from oracle import pe_oracle as O
    levels = np.linspace(-60.0, 6.0, 133)
    for mode in ("compress", "limit", "expand", "gate"):
        for knee in (0.0, 7.0):
            d = DynamicsPE(ConstantPE(1.0), ConstantPE(1.0), threshold=-18.0, ratio=3.0, knee=knee, makeup_gain=0.0,
                           mode=DynamicsMode(mode), gate_range=-50.0)
            want = O.dynamics_gain_db(levels, -18.0, 3.0, knee, mode, -50.0)
            got = np.array([d._compute_gain_db(float(v)) for v in levels])
            assert np.allclose(got, want, rtol=0, atol=1e-12), (mode, knee)


def test_compressor_family_properties():
    src = SinePE(frequency=440.0)
    c = CompressorPE(src)
    assert (c.threshold, c.ratio, c.attack, c.release, c.knee, c.lookahead, c.detection, c.stereo_link) == (
        -20.0, 4.0, 0.01, 0.1, 6.0, 0.0, DetectionMode.RMS, True)
    c = CompressorPE(src, threshold=-15.0, ratio=8.0, attack=0.005, release=0.2, knee=12.0, makeup_gain=6.0,
                     lookahead=0.003, detection=DetectionMode.PEAK, stereo_link=False)
    assert (c.threshold, c.ratio, c.attack, c.release, c.knee, c.makeup_gain, c.lookahead, c.detection,
            c.stereo_link) == (-15.0, 8.0, 0.005, 0.2, 12.0, 6.0, 0.003, DetectionMode.PEAK, False)
    c = CompressorPE(src)
    assert len(c.inputs()) == 1 and isinstance(c.inputs()[0], DynamicsPE) and c.is_pure() is False
    assert CompressorPE(ConstantPE(1.0, channels=2)).channel_count() == 2
    assert CompressorPE(src, threshold=-20, ratio=4, makeup_gain="auto").makeup_gain > 0
    assert all(p in repr(CompressorPE(src, threshold=-20, ratio=4)) for p in ("CompressorPE", "threshold=-20", "ratio=4"))
    lim = LimiterPE(src)
    assert (lim.ceiling, lim.release, lim.lookahead) == (-1.0, 0.05, 0.005)
    lim = LimiterPE(src, ceiling=-3.0, release=0.1, lookahead=0.01)
    assert (lim.ceiling, lim.release, lim.lookahead) == (-3.0, 0.1, 0.01)
    assert "LimiterPE" in repr(lim) and "ceiling=-3.0" in repr(lim)
    gate = ExpanderPE(src)
    assert (gate.threshold, gate.attack, gate.release, gate.gate_range) == (-40.0, 0.001, 0.05, -80.0)
    gate = ExpanderPE(src, threshold=-30.0, attack=0.0005, release=0.1, gate_range=-60.0)
    assert (gate.threshold, gate.attack, gate.release, gate.gate_range) == (-30.0, 0.0005, 0.1, -60.0)
    assert gate.is_pure() is False and "ExpanderPE" in repr(gate)
