"""GPU: LoopPE / WindowPE / DynamicsPE / CompressorPE / LimiterPE / ExpanderPE rendered through the HIP path:
the behaviour the reference's own tests assert (test_loop_pe.py:126-358, test_window_pe.py:96-375,
test_dynamics_pe.py:126-434, test_compressor_pe.py:114-487), and direct oracle comparisons at sizes and window
lengths the golden cases do not reach."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SR = 44100


@pytest.fixture
def pg():
    import pygmu2_amd as pg
    pg.set_sample_rate(SR)
    return pg


def _render(pg, pe, start, n):
    r = pg.NullRenderer(sample_rate=SR)
    r.set_source(pe)
    r.start()
    out = pe.render(start, n).data
    r.stop()
    return out


def test_loop_behaviour(pg):
    ramp = pg.PiecewisePE([(0, 0.0), (100, 1.0)])
    loop = pg.LoopPE(ramp)
    first, second = _render(pg, loop, 0, 100), _render(pg, loop, 100, 100)
    assert np.array_equal(first, _render(pg, ramp, 0, 100)) and np.array_equal(first, second)
    wrap = _render(pg, pg.LoopPE(pg.PiecewisePE([(0, 0.0), (5, 4.0)])), 3, 4)[:, 0]
    assert wrap == pytest.approx([2.4, 3.2, 0.0, 0.8], abs=1e-6)
    region = _render(pg, pg.LoopPE(pg.PiecewisePE([(0, 0.0), (10, 9.0)]), loop_start=2, loop_end=5), 0, 9)[:, 0]
    assert region == pytest.approx([1.8, 2.7, 3.6] * 3, abs=1e-6)
    counted = _render(pg, pg.LoopPE(pg.ConstantPE(1.0), loop_start=0, loop_end=100, count=2), 150, 100)[:, 0]
    assert np.all(counted[:50] == 1.0) and np.all(counted[50:] == 0.0)          # partial final render
    assert np.all(_render(pg, pg.LoopPE(pg.ConstantPE(1.0), loop_end=100, count=2), 200, 64) == 0.0)
    stereo = pg.ArrayPE(np.stack([np.arange(50), -np.arange(50)], axis=1).astype(np.float32))
    out = _render(pg, pg.LoopPE(stereo), 40, 30)
    assert np.array_equal(out[:, 0], (np.arange(40, 70) % 50).astype(np.float32)) and np.array_equal(out[:, 1], -out[:, 0])


def test_loop_crossfade_softens_the_seam(pg):
    """test_loop_pe.py:277-330 (at 1 kHz there: 20-frame fade over a 100-frame ramp): the end of the loop fades into
    its first frames, so the step at the loop point shrinks from ~1 to the height the ramp reaches in one fade."""
    pg.set_sample_rate(1000)
    saw = pg.PiecewisePE([(0, 0.0), (100, 1.0)])
    plain = pg.LoopPE(saw, crossfade_seconds=0.0)
    faded = pg.LoopPE(saw, crossfade_seconds=0.02)
    assert faded.crossfade_samples == 20

    def run(pe, start, n):
        r = pg.NullRenderer(sample_rate=1000)
        r.set_source(pe)
        r.start()
        out = pe.render(start, n).data[:, 0]
        r.stop()
        return out
    hard = run(plain, 99, 2)
    assert hard[0] > 0.9 and hard[1] < 0.1
    assert np.max(np.abs(np.diff(run(faded, 80, 40)))) < 0.3
    pg.set_sample_rate(SR)


@pytest.mark.parametrize("mode", ["max", "min", "mean", "rms"])
@pytest.mark.parametrize("window", [0.0, 0.0005, 0.05, 0.6])
def test_window_matches_oracle(pg, mode, window):
    """Window lengths from 3 frames to 26 461 (more than the render itself), stereo, a negative start."""
    from oracle import pe_oracle as O
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((30000, 2)) * 0.4).astype(np.float32)
    pe = pg.WindowPE(pg.ArrayPE(x), window=window, mode=pg.WindowMode(mode))
    half = O.window_half(window, SR)
    for start, n in ((-700, 5000), (12000, 20000)):
        got = _render(pg, pe, start, n)
        idx = np.arange(start - half, start + n + half)
        padded = np.where(((idx >= 0) & (idx < len(x)))[:, None], x[np.clip(idx, 0, len(x) - 1)], 0.0).astype(np.float32)
        want = O.window_stat(padded, n, half, mode)
        if mode in ("max", "min"):
            assert np.array_equal(got, want), (mode, window, start)
        else:
            assert np.max(np.abs(got.astype(np.float64) - want)) <= 1e-5 * np.max(np.abs(want)) + 1e-7


def test_window_behaviour(pg):
    const = _render(pg, pg.WindowPE(pg.ConstantPE(0.5), window=0.01, mode=pg.WindowMode.MAX), 0, 100)
    assert np.all(const == 0.5)
    peak = _render(pg, pg.WindowPE(pg.DiracPE(), window=0.023, mode=pg.WindowMode.MAX), 0, 1000)[:, 0]
    half = int(0.023 * SR / 2)
    assert np.all(peak[:half + 1] == 1.0) and np.all(peak[half + 1:] == 0.0)
    mean = _render(pg, pg.WindowPE(pg.ConstantPE(-0.7), window=0.01, mode=pg.WindowMode.MEAN, rectify=False), 0, 100)
    assert mean == pytest.approx(-0.7, abs=1e-6)
    rms = _render(pg, pg.WindowPE(pg.SinePE(frequency=441.0), window=0.1, mode=pg.WindowMode.RMS), 10000, 2000)
    assert rms == pytest.approx(1.0 / np.sqrt(2.0), abs=1e-3)
    a = _render(pg, pg.WindowPE(pg.SinePE(frequency=100.0), window=0.01), 500, 300)
    b = _render(pg, pg.WindowPE(pg.SinePE(frequency=100.0), window=0.01), 600, 200)
    assert np.array_equal(a[100:], b)                                   # render-order independent (pure)


def test_dynamics_behaviour(pg):
    def run(level, **kw):
        d = pg.DynamicsPE(pg.ConstantPE(1.0), pg.ConstantPE(level), makeup_gain=0.0, **kw)
        return float(_render(pg, d, 0, 64)[0, 0])
    assert run(0.05, threshold=-20.0, ratio=4.0) == pytest.approx(1.0, abs=1e-6)          # -26 dB: below threshold
    assert run(1.0, threshold=-20.0, ratio=4.0) == pytest.approx(10 ** (-15.0 / 20.0), rel=1e-5)
    assert run(1.0, threshold=-20.0, ratio=8.0) < run(1.0, threshold=-20.0, ratio=2.0)
    assert run(1.0, threshold=-6.0, mode=pg.DynamicsMode.LIMIT) == pytest.approx(10 ** (-6.0 / 20.0), rel=1e-5)
    assert run(0.001, threshold=-40.0, mode=pg.DynamicsMode.GATE) == pytest.approx(1e-4, rel=1e-5)
    assert run(0.5, threshold=-40.0, mode=pg.DynamicsMode.GATE) == 1.0
    assert run(0.01, threshold=-20.0, ratio=2.0, mode=pg.DynamicsMode.EXPAND) == pytest.approx(10 ** (-20.0 / 20.0), rel=1e-5)
    assert run(0.5, threshold=-20.0, ratio=2.0, mode=pg.DynamicsMode.EXPAND) == 1.0
    soft = [run(10 ** (db / 20.0), threshold=-20.0, ratio=4.0, knee=12.0) for db in (-30.0, -22.0, -18.0, -10.0)]
    assert soft[0] == pytest.approx(1.0, abs=1e-6) and soft[0] > soft[1] > soft[2] > soft[3]
    # stereo link: the louder envelope channel drives both audio channels; unlinked: each its own
    env = pg.ArrayPE(np.tile(np.array([[1.0, 0.01]], dtype=np.float32), (64, 1)), extend_mode=pg.ExtendMode.HOLD_BOTH)
    linked = _render(pg, pg.DynamicsPE(pg.ConstantPE(1.0, channels=2), env, threshold=-20.0, ratio=4.0, makeup_gain=0.0), 0, 64)
    split = _render(pg, pg.DynamicsPE(pg.ConstantPE(1.0, channels=2), env, threshold=-20.0, ratio=4.0, makeup_gain=0.0,
                                      stereo_link=False), 0, 64)
    assert linked[0, 0] == linked[0, 1] < 1.0 and split[0, 0] == linked[0, 0] and split[0, 1] == 1.0


def test_compressor_family_behaviour(pg):
    loud = _render(pg, pg.CompressorPE(pg.ConstantPE(1.0), threshold=-20.0, ratio=4.0, makeup_gain=0.0), 0, 20000)
    assert 0.15 < float(loud[-1, 0]) < 0.25                              # 20 dB over at 4:1 -> -15 dB
    quiet = _render(pg, pg.CompressorPE(pg.ConstantPE(0.01), threshold=-20.0, ratio=4.0, makeup_gain=0.0), 0, 5000)
    assert quiet[-1, 0] == pytest.approx(0.01, rel=1e-4)
    lim = _render(pg, pg.LimiterPE(pg.ConstantPE(1.0), ceiling=-6.0), 0, 2000)
    assert np.all(lim[1000:, 0] < 0.6)
    gate_quiet = _render(pg, pg.ExpanderPE(pg.ConstantPE(0.001), threshold=-40.0), 0, 4000)
    gate_loud = _render(pg, pg.ExpanderPE(pg.ConstantPE(0.5), threshold=-40.0), 0, 4000)
    assert np.all(np.abs(gate_quiet[2000:]) < 1e-6) and gate_loud[-1, 0] == pytest.approx(0.5, rel=1e-4)
    # state persists across renders, and a restart clears it (test_compressor_pe.py:169-230)
    # (peak detection: the RMS detector is block-local in the reference -- its window clamps at the block's edges)
    comp = pg.CompressorPE(pg.SinePE(frequency=440.0), threshold=-20.0, detection=pg.DetectionMode.PEAK)
    r = pg.NullRenderer(sample_rate=SR)
    r.set_source(comp)
    r.start()
    whole = np.concatenate([comp.render(0, 3000).data, comp.render(3000, 3000).data])
    r.stop()
    r.start()
    again = comp.render(0, 6000).data
    r.stop()
    assert np.max(np.abs(whole - again)) <= 1e-5 * np.max(np.abs(again))
