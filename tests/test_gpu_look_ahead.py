"""GPU: look-ahead of stateful sub-graphs (pygmu2_amd/look_ahead.py).  Whatever the caller does between blocks --
keep streaming, seek, change the block size, reset or pull a PE inside the graph, stop and restart -- samples
and carried states are the ones block-by-block rendering produces (within the filters' partition tolerance;
exactly for the bit-exact PEs), and both agree with the CPU oracle."""

import numpy as np
import pytest

import pygmu2_amd as pg
from pygmu2_amd import look_ahead, transforms as tf

pytestmark = pytest.mark.gpu

SR = 44100


def autowah(kind="biquad"):
    src = pg.SinePE(frequency=220.0, amplitude=0.8)
    env = pg.EnvelopePE(src, attack=0.005, release=0.05, mode=pg.DetectionMode.PEAK)
    ctl = pg.TransformPE(env, func=tf.Chain(tf.Clip(0.0, 1.0), tf.Sqrt(), tf.Affine(2900.0, 100.0)), name="ctl")
    flt = (pg.BiquadPE if kind == "biquad" else pg.SVFilterPE)(src, frequency=ctl, q=10.0, mode=pg.BiquadMode.LOWPASS)
    root = pg.CropPE(pg.GainPE(flt, gain=1.0), 0, 8 * SR)
    return root, {"filter": flt, "env": env}


def c2():
    flt = pg.BiquadPE(pg.SinePE(frequency=440.0), frequency=1000.0, q=0.707)
    return flt, {"filter": flt}


def voice():
    osc = pg.BlitSawPE(110.0)
    flt = pg.BiquadPE(osc, frequency=2000.0, q=0.707)
    env = pg.AdsrGatedPE(pg.PeriodicGate(7.0, 0.5), 0.01, 0.02, 0.7, 0.03)
    root = pg.GainPE(flt, gain=env)
    return root, {"filter": flt, "osc": osc, "env": env}


def comb_ladder():
    osc = pg.SuperSawPE(98.0, voices=3, seed=5)
    lad = pg.LadderPE(osc, frequency=1200.0, resonance=0.3, oversample=2)
    comb = pg.CombPE(lad, frequency=440.0, feedback=0.6)
    return comb, {"filter": lad, "osc": osc, "comb": comb}


def conv():
    h = (np.random.default_rng(1).standard_normal(300) * np.exp(-np.arange(300) / 60.0)).astype(np.float32)
    c = pg.ConvolvePE(pg.SinePE(330.0, channels=2), pg.ArrayPE(h))
    return c, {"filter": c}


GRAPHS = {"autowah_biquad": lambda: autowah("biquad"), "autowah_svf": lambda: autowah("svf"), "c2": c2,
          "voice": voice, "comb_ladder": comb_ladder, "conv": conv}
TOL = {"conv": 5e-6}          # the float32 MFMA accumulation groups its taps by block


def run(make, script, ahead):
    """script: ("r", start, n) render | ("reset", name) | ("inner", name, start, n) | ("restart",)."""
    look_ahead.set_enabled(ahead)
    try:
        pg.set_sample_rate(SR)
        root, named = make()
        r = pg.NullRenderer(SR)
        r.set_source(root)
        r.start()
        out = []
        for step in script:
            if step[0] == "r":
                out.append(root.render(step[1], step[2]).data.copy())
            elif step[0] == "reset":
                named[step[1]].reset_state()
            elif step[0] == "inner":
                out.append(named[step[1]].render(step[2], step[3]).data.copy())
            elif step[0] == "restart":
                r.stop()
                r.start()
        r.stop()
        return out, root
    finally:
        look_ahead.set_enabled(True)


def close(a, b, tol=2e-6):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.shape == y.shape, i
        peak = max(float(np.max(np.abs(y))), 1e-3)
        err = float(np.max(np.abs(x.astype(np.float64) - y)))
        assert err <= tol * peak, (i, err, peak)


STREAM = [("r", i * 1024, 1024) for i in range(150)]
SCRIPTS = {
    "stream": STREAM,
    "seek_back_and_forth": STREAM[:10] + [("r", 3000, 512), ("r", 3512, 512), ("r", 4024, 512), ("r", 100000, 777),
                                           ("r", 100777, 777), ("r", 101554, 777), ("r", 0, 64)],
    "ragged_blocks": [("r", s, n) for s, n in zip(np.cumsum([0] + [17, 23, 19, 41, 7, 93] * 40)[:-1],
                                                  [17, 23, 19, 41, 7, 93] * 40)],
    "reset_inside": STREAM[:7] + [("reset", "filter")] + [("r", (7 + i) * 1024, 1024) for i in range(70)],
    "inner_pull": STREAM[:5] + [("inner", "filter", 5 * 1024, 300)] + [("r", 5 * 1024 + 300 + i * 1024, 1024)
                                                                     for i in range(5)],
    "restart": STREAM[:9] + [("restart",)] + STREAM[:9],
    "block_44100": [("r", i * 44100, 44100) for i in range(14)] + [("r", 5, 44100)],
}


@pytest.mark.parametrize("script", sorted(SCRIPTS))
@pytest.mark.parametrize("graph", sorted(GRAPHS))
def test_look_ahead_is_invisible(graph, script):
    if graph == "conv" and script == "inner_pull":
        pytest.skip("the ConvolvePE is the root of that graph")
    steps = SCRIPTS[script]
    got, root = run(GRAPHS[graph], steps, ahead=True)
    assert look_ahead.capable(root), "the graph was expected to take the look-ahead path"
    want, _ = run(GRAPHS[graph], steps, ahead=False)
    close(got, want, TOL.get(graph, 2e-6))


def test_streams_really_come_from_windows_and_states_roll_back():
    pg.set_sample_rate(SR)
    flt, _ = c2()
    r = pg.NullRenderer(SR)
    r.set_source(flt)
    r.start()
    for i in range(5):
        flt.render(i * 1024, 1024)
    win = flt.__dict__.get("_la_win")
    assert win is not None and win.first == 1024 and win.end == 1024 + look_ahead.FIRST_WINDOW_BLOCKS * 1024
    assert win.served == 5 * 1024
    for i in range(5, 30):                                    # slow start: every refill is WINDOW_GROWTH times longer
        flt.render(i * 1024, 1024)
    win = flt.__dict__.get("_la_win")
    first, length = 1, look_ahead.FIRST_WINDOW_BLOCKS
    while first + length <= 29:                               # the window that holds block 29
        first, length = first + length, min(length * look_ahead.WINDOW_GROWTH, look_ahead.AHEAD_BLOCKS)
    assert win.first == first * 1024 and win.end == (first + length) * 1024 and win.served == 30 * 1024
    ahead_state = flt._state.to_host().copy()             # the state at the end of the window
    look_ahead.settle(flt)
    assert "_la_win" not in flt.__dict__
    settled = flt._state.to_host().copy()
    r.stop()
    ref, _ = c2()
    look_ahead.set_enabled(False)
    try:
        r2 = pg.NullRenderer(SR)
        r2.set_source(ref)
        r2.start()
        for i in range(30):
            ref.render(i * 1024, 1024)
        want = ref._state.to_host().copy()
        r2.stop()
    finally:
        look_ahead.set_enabled(True)
    assert not np.allclose(ahead_state, want)
    assert np.allclose(settled, want, rtol=1e-9, atol=1e-12)


def test_graphs_that_must_not_look_ahead():
    pg.set_sample_rate(SR)
    src = pg.SinePE(220.0)
    assert not look_ahead.capable(pg.GainPE(src, 0.5))                               # pure: read-ahead's business
    assert look_ahead.capable(pg.EnvelopePE(src, mode=pg.DetectionMode.RMS))         # block-local RMS: told the period
    assert not look_ahead.capable(pg.TransformPE(pg.BiquadPE(src, 500.0, 1.0), func=lambda x: x * 2.0))
    assert look_ahead.capable(pg.CompressorPE(src))                                  # CachePE over a pure source
    fm = pg.SinePE(frequency=pg.GainPE(pg.SinePE(3.0), 50.0), phase=0.5)           # the reference re-adds the offset
    assert not look_ahead.capable(fm) and look_ahead.capable(pg.SinePE(frequency=pg.GainPE(pg.SinePE(3.0), 50.0)))
    bank = pg.MixPE(*[pg.BlitSawPE(100.0 + i) for i in range(6)])
    assert not look_ahead.capable(bank)                                              # voice bank keeps its own states
    assert look_ahead.capable(pg.MixPE(pg.BlitSawPE(100.0), pg.BlitSawPE(150.0)))


def test_autowah_matches_the_oracle_through_look_ahead():
    from oracle import graph_eval
    from oracle.golden_cases import S
    pg.set_sample_rate(SR)
    root, _ = autowah("biquad")
    r = pg.NullRenderer(SR)
    r.set_source(root)
    r.start()
    got = np.concatenate([root.render(i * 1024, 1024).data for i in range(40)])
    r.stop()
    src = S("SinePE", frequency=220.0, amplitude=0.8)
    env = S("EnvelopePE", source=src, attack=0.005, release=0.05, mode="peak")
    ctl = S("TransformPE", source=env, ops=[["clip", 0.0, 1.0], ["sqrt"], ["affine", 2900.0, 100.0]])
    g = graph_eval.Node(S("GainPE", source=S("BiquadPE", source=src, frequency=ctl, q=10.0, mode="lowpass"),
                          gain=1.0), SR)
    want = np.concatenate([g.render(i * 1024, 1024) for i in range(40)])
    assert np.max(np.abs(got - want)) <= 1e-5 * np.max(np.abs(want))


def _compressor():
    src = pg.MixPE(pg.SinePE(330.0), pg.SinePE(331.3))                       # beating: a level that moves (pure)
    c = pg.CompressorPE(pg.GainPE(src, 0.7), threshold=-18.0, ratio=4.0, attack=0.004, release=0.06)
    return c, {"filter": c}


def _limiter():
    lim = pg.LimiterPE(pg.GainPE(pg.SinePE(220.0), 1.4), ceiling=-3.0)
    return lim, {"filter": lim}


@pytest.mark.parametrize("script", ["stream", "seek_back_and_forth", "restart", "block_44100"])
@pytest.mark.parametrize("graph", ["compressor", "limiter"])
def test_side_chain_processors_stream_through_windows(graph, script):
    """CompressorPE's RMS detector restarts at every block edge in the reference; a window renders many blocks at
    once and tells the detector where the caller's edges are.  LimiterPE: peak detection with look-ahead."""
    make = {"compressor": _compressor, "limiter": _limiter}[graph]
    steps = SCRIPTS[script]
    got, root = run(make, steps, ahead=True)
    assert look_ahead.capable(root)
    want, _ = run(make, steps, ahead=False)
    close(got, want, 2e-6)


def test_block_sensitive_graphs_need_blocks_passed_through():
    pg.set_sample_rate(SR)
    comp = pg.CompressorPE(pg.SinePE(300.0))
    assert look_ahead.capable(comp) and comp.__dict__["_la_sensitive"]
    cropped = pg.CropPE(pg.CompressorPE(pg.SinePE(300.0)), 100, 50_000)      # CropPE re-cuts the blocks it pulls
    assert not look_ahead.capable(cropped)
    stateful_src = pg.CompressorPE(pg.BiquadPE(pg.SinePE(300.0), 800.0, 1.0))     # the CachePE would show
    assert not look_ahead.capable(stateful_src)
    # a window of a block-sensitive graph serves only the block size it was opened with
    r = pg.NullRenderer(SR); r.set_source(comp); r.start()
    for i in range(4):
        comp.render(i * 1024, 1024)
    assert comp.__dict__["_la_win"].block == 1024
    comp.render(4096, 500)                                   # another block size: the old window is settled ...
    comp.render(4596, 500)
    win = comp.__dict__["_la_win"]
    assert win.block == 500 and win.first in (4096, 4596)    # ... and the stream goes on in windows of the new one
    r.stop()


def test_window_that_never_pulls_the_fused_filter_keeps_its_state():
    """CropPE(BiquadPE(SinePE)) streamed past the crop's end.  The window rendered past the end returns fill
    without pulling the filter, whose snapshot copy was to be written by the filter's own next kernel: look_ahead makes
    that copy after the window's render, so a seek back into the crop filters from the carried state, not from
    uninitialised pool memory."""
    def crop():
        flt = pg.BiquadPE(pg.SinePE(frequency=440.0), frequency=1000.0, q=0.707)
        root = pg.CropPE(flt, 0, 20 * 1024)
        return root, {"filter": flt}
    steps = ([("r", i * 1024, 1024) for i in range(20)]                  # the crop's own frames (windows of 8, 16)
             + [("r", (20 + i) * 1024, 1024) for i in range(30)]         # past the end: fill, new windows, no pulls
             + [("inner", "filter", 20 * 1024, 2048)]                    # mid-window: settle, then the filter itself
             + [("r", 5 * 1024, 1024), ("r", 6 * 1024, 1024)])           # and a seek back into the crop
    got, root = run(crop, steps, ahead=True)
    want, _ = run(crop, steps, ahead=False)
    assert all(np.all(np.isfinite(g)) for g in got)
    close(got, want)


def test_unexpected_errors_inside_a_window_are_not_swallowed():
    """A window render may fail for reasons of its size (declined, out of memory, a kernel's size limit): those fall
    back to block by block.  Anything else reaches the caller."""
    pg.set_sample_rate(SR)
    flt, _ = c2()
    r = pg.NullRenderer(SR)
    r.set_source(flt)
    r.start()
    flt.render(0, 1024)
    real = flt._render
    calls = {"n": 0}

    def boom(start, duration):
        calls["n"] += 1
        if duration > 1024:
            raise ZeroDivisionError("a bug, not a size limit")
        return real(start, duration)
    flt._render = boom
    with pytest.raises(ZeroDivisionError):
        flt.render(1024, 1024)
    assert flt.__dict__.get("_la_ok") is False

    flt2, _ = c2()
    r2 = pg.NullRenderer(SR)
    r2.set_source(flt2)
    r2.start()
    flt2.render(0, 1024)
    real2 = flt2._render

    def too_big(start, duration):
        if duration > 1024:
            raise MemoryError("no room for the window")
        return real2(start, duration)
    flt2._render = too_big
    got = flt2.render(1024, 1024).data.copy()                # falls back to the block itself
    flt2._render = real2
    ref, _ = c2()
    r3 = pg.NullRenderer(SR)
    r3.set_source(ref)
    r3.start()
    look_ahead.set_enabled(False)
    try:
        ref.render(0, 1024)
        want = ref.render(1024, 1024).data.copy()
    finally:
        look_ahead.set_enabled(True)
    close([got], [want])
