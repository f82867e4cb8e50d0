"""
GPU: the time-segmented LadderPE path (pgx_ladder with settle_frames > 0) against the
sequential kernel and the oracle (orc_ladder, the restated numba kernel of ladder_pe.py:31-203).

The segmented path is only a speed device: the library checks every warm-started segment and
re-renders the chain sequentially when the check fails.  Both outcomes are exercised here.
"""

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


@pytest.fixture(scope="module")
def env():
    import pygmu2_amd as pg
    from pygmu2_amd import device
    lib = device.ensure_init()

    class Env:
        pass

    e = Env()
    e.pg, e.device, e.lib = pg, device, lib
    return e


def _ladder(e, x, settle, *, freq=1200.0, res=0.3, drive=1.0, mode=0, oversample=2, sr=48000.0, state0=None,
            accurate=0):
    device, lib = e.device, e.lib
    n, ch = x.shape
    xin = device.DeviceBuffer.from_host(x)
    out = device.DeviceBuffer((n, ch), np.float32)
    params = device.upload_struct(device.LADDER_PARAMS, freq=freq, resonance=res, drive=drive, passband_gain=0.5,
                                  oversample=oversample, mode=mode)
    st = device.DeviceBuffer.from_host(np.zeros((ch, 9)) if state0 is None else np.asarray(state0, dtype=np.float64))
    need = lib.pgx_ladder_workspace_bytes(1, n, ch, settle)
    ws = device.DeviceBuffer((max(need, 8),), np.uint8, zero=True)
    device.check(lib.pgx_ladder(out.ptr, 0, xin.ptr, 0, 1, n, ch, sr, params.ptr, None, None, None, st.ptr,
                                settle, accurate, ws.ptr if need else None))
    fallbacks = int(ws.to_host()[0:4].view(np.int32)[0]) if need else -1
    return out.to_host(), st.to_host(), fallbacks, need


def _signal(n, ch, seed, silence=False):
    rng = np.random.default_rng(seed)
    t = np.arange(n)[:, None] / 48000.0
    x = 0.6 * np.sin(2 * np.pi * 110.0 * t * (1 + np.arange(ch))) + 0.2 * rng.standard_normal((n, ch))
    if silence:
        x[n // 3: n // 3 + 5000] = 0.0                       # exercises the |in| < 1e-5 state-decay rule
    return x.astype(np.float32)


@pytest.mark.parametrize("mode", [0, 2, 4])
@pytest.mark.parametrize("res,freq", [(0.0, 1200.0), (0.3, 1200.0), (0.4, 3000.0)])
def test_segmented_matches_sequential(env, mode, res, freq):
    from pygmu2_amd.ladder_pe import ladder_settle_frames
    n = 48_000
    x = _signal(n, 2, 3, silence=True)
    settle = ladder_settle_frames(freq, res, 48000.0, 2)
    assert settle > 0
    s0 = np.random.default_rng(1).standard_normal((2, 9)) * 0.05
    y_seq, st_seq, _, need0 = _ladder(env, x, 0, freq=freq, res=res, mode=mode, state0=s0)
    y_seg, st_seg, fallbacks, need = _ladder(env, x, settle, freq=freq, res=res, mode=mode, state0=s0)
    assert need0 == 0 and need > 0
    assert fallbacks == 0, "the host's settle estimate should hold for this (stable) setting"
    peak = float(np.max(np.abs(y_seq)))
    assert float(np.max(np.abs(y_seg.astype(np.float64) - y_seq))) <= 1e-6 * peak
    assert np.allclose(st_seg, st_seq, rtol=1e-6, atol=1e-9)
    # the two-speed warm-up LadderPE really uses: float32 tanh first, the float64 one for the tail
    accurate = ladder_settle_frames(freq, res, 48000.0, 2, target=4e-4)
    assert 0 < accurate < settle
    y_two, st_two, fallbacks, _ = _ladder(env, x, settle, freq=freq, res=res, mode=mode, state0=s0, accurate=accurate)
    assert fallbacks == 0, "the fast part of the warm-up left more than the accurate tail could contract"
    assert float(np.max(np.abs(y_two.astype(np.float64) - y_seq))) <= 1e-6 * peak
    assert np.allclose(st_two, st_seq, rtol=1e-6, atol=1e-9)


def test_failed_check_rerenders_sequentially(env):
    """A hopeless warm-up length (self-oscillating filter, 64 samples) must fall back and still be exact."""
    n = 20_000
    x = _signal(n, 1, 4)
    y_seq, st_seq, _, _ = _ladder(env, x, 0, res=0.95, freq=2000.0)
    y_seg, st_seg, fallbacks, need = _ladder(env, x, 64, res=0.95, freq=2000.0)
    assert need > 0 and fallbacks == 1
    assert np.array_equal(y_seg, y_seq)
    assert np.array_equal(st_seg, st_seq)


def test_segmented_matches_oracle_and_streams_state(env):
    """LadderPE over 3 x 48000-frame blocks (segmented each) == one oracle pass."""
    from oracle import pe_oracle as O
    pg = env.pg
    pg.set_sample_rate(48000)
    x = _signal(144_000, 1, 5)
    pe = pg.LadderPE(pg.ArrayPE(x), frequency=1200.0, resonance=0.3, mode=pg.LadderMode.LP24, oversample=2)
    assert pe._settle_frames() > 0
    r = pg.NullRenderer(sample_rate=48000)
    r.set_source(pe)
    r.start()
    got = np.concatenate([pe.render(i * 48_000, 48_000).data for i in range(3)])
    r.stop()
    st = O.ladder_state(1)
    want = O.ladder(st, x, 1200.0, resonance=0.3, mode="lp24", drive=1.0, oversample=2, sr=48000)
    peak = float(np.max(np.abs(want)))
    assert float(np.max(np.abs(got.astype(np.float64) - want))) <= REL_TOL * peak


def test_pe_driven_cutoff_and_resonance_run_in_time_segments():
    """LadderPE with a PE cutoff (and resonance): the warm-up length of the time segments comes from the block's
    lowest cutoff / highest resonance (pgx_stream_range); against the sequential kernel and the oracle, and the
    device's own check must not have fallen back."""
    import pygmu2_amd as pg
    from pygmu2_amd import ladder_pe, look_ahead
    from oracle import graph_eval
    from oracle.golden_cases import S
    pg.set_sample_rate(48000)
    sine = lambda f, a: S("SinePE", frequency=f, amplitude=a)
    spec = S("LadderPE", source=S("BlitSawPE", frequency=110.0),
             frequency=S("MixPE", inputs=[S("ConstantPE", value=1500.0), sine(0.7, 900.0)]),
             resonance=S("MixPE", inputs=[S("ConstantPE", value=0.3), sine(0.31, 0.2)]),
             mode="lp24", drive=1.0, oversample=2)
    blocks = [(0, 48_000), (48_000, 48_000), (96_000, 20_000), (116_000, 4096)]

    def run(segmented):
        import spec_build
        keep = ladder_pe.SEGMENT_STREAM_LADDER
        ladder_pe.SEGMENT_STREAM_LADDER = segmented
        look_ahead.set_enabled(False)
        try:
            pe = spec_build.build(spec)
            r = pg.NullRenderer(sample_rate=48000)
            r.set_source(pe)
            r.start()
            outs = [pe.render(s, n).data.copy() for s, n in blocks]
            r.stop()
            return outs, pe
        finally:
            ladder_pe.SEGMENT_STREAM_LADDER = keep
            look_ahead.set_enabled(True)

    seg, pe = run(True)
    assert pe._stream_settle_cache and all(v[0] > 0 for v in pe._stream_settle_cache.values())
    seq, _ = run(False)
    g = graph_eval.Node(spec, 48000)
    for (s, n), a, b in zip(blocks, seg, seq):
        w = g.render(s, n)
        peak = float(np.max(np.abs(w)))
        assert float(np.max(np.abs(a.astype(np.float64) - b))) <= 1e-6 * peak
        assert float(np.max(np.abs(a.astype(np.float64) - w))) <= 1e-5 * peak


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PGX_FUZZ_LADDER", "12"))))
def test_random_ladders_with_streams_against_the_oracle(seed):
    """Random LadderPEs -- every mode, oversample 1..4, scalar or PE-driven cutoff / resonance / drive (sweeps of random
    depth and rate), mono or stereo oscillator in front, pulls from 17 frames to 150 000 (sequential kernel, time
    segments with the warm-up taken from the block's own range, look-ahead windows over streams of small blocks) --
    against the oracle's loop.  Resonance stays below 0.6: the region where the reference is well-conditioned."""
    import pygmu2_amd as pg
    import spec_build
    from oracle import graph_eval
    from oracle.golden_cases import S
    rng = np.random.default_rng(88_000 + seed)
    sr = int(rng.choice([44100, 48000]))
    ch = int(rng.choice([1, 2]))
    sine = lambda f, a: S("SinePE", frequency=f, amplitude=a)

    def maybe_stream(lo, hi):
        mid, dev = 0.5 * (lo + hi), 0.5 * (hi - lo)
        if rng.random() < 0.5:
            return float(rng.uniform(lo, hi))
        return S("MixPE", inputs=[S("ConstantPE", value=mid), sine(float(rng.uniform(0.1, 6.0)), float(rng.uniform(0.1, 1.0)) * dev)])

    src = (S("BlitSawPE", frequency=float(rng.uniform(40.0, 800.0)), channels=ch) if rng.random() < 0.6
           else S("SuperSawPE", frequency=float(rng.uniform(40.0, 400.0)), voices=int(rng.integers(2, 6)), seed=int(seed),
                  channels=ch))
    spec = S("LadderPE", source=src, frequency=maybe_stream(150.0, 5000.0), resonance=maybe_stream(0.0, 0.6),
             mode=str(rng.choice(["lp24", "lp12", "bp12", "hp24", "hp12"])), drive=maybe_stream(0.5, 2.0),
             oversample=int(rng.integers(1, 5)))
    pattern = int(rng.integers(0, 3))
    if pattern == 0:
        sizes = [int(rng.choice([17, 1000, 5000, 20_000])) for _ in range(4)]
    elif pattern == 1:
        sizes = [int(rng.choice([48_000, 100_001, 150_000])), 4096, int(rng.choice([17, 48_000]))]
    else:
        sizes = [int(rng.choice([1024, 4096]))] * int(rng.integers(12, 30))
    blocks, pos = [], 0
    for n in sizes:
        blocks.append((pos, n))
        pos += n
    pg.set_sample_rate(sr)
    pe = spec_build.build(spec)
    r = pg.NullRenderer(sample_rate=sr)
    r.set_source(pe)
    r.start()
    got = [pe.render(s, n).data.copy() for s, n in blocks]
    r.stop()
    g = graph_eval.Node(spec, sr)
    want = [g.render(s, n) for s, n in blocks]
    peak = max(float(np.max(np.abs(w))) for w in want) or 1.0
    for i, ((s, n), a, w) in enumerate(zip(blocks, got, want)):
        err = float(np.max(np.abs(a.astype(np.float64) - w)))
        assert err <= 1e-5 * peak, (i, s, n, err, peak, sr, spec)


def test_segment_samples_on_fused_arithmetic_against_the_reference_order(tmp_path):
    """A segment's own samples run on the warm-up's fused, regrouped stages and a tanh good to ~1e-11 (ladder_emit);
    PGX_LADDER_EXACT_EMIT=1 keeps the reference's operation order and pgx_tanh for them.  Same graph, same blocks, two
    processes (the switch is read once): the two agree to 2e-7 of the peak -- a float32 ulp or two -- for every mode."""
    import os
    import subprocess
    import sys
    script = tmp_path / "w.py"
    script.write_text(r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PGX_ROOT"])
import pygmu2_amd as pg
from pygmu2_amd import look_ahead
pg.set_sample_rate(48000)
look_ahead.set_enabled(False)
outs = []
for mode in ("lp24", "lp12", "bp12", "hp24", "hp12"):
    pe = pg.LadderPE(pg.SuperSawPE(110.0, voices=5, seed=3, channels=2), frequency=1500.0, resonance=0.45,
                     mode=pg.LadderMode(mode), drive=1.3, oversample=2)
    r = pg.NullRenderer(48000); r.set_source(pe); r.start()
    outs += [pe.render(i * 96_000, 96_000).data.copy() for i in range(2)]
    assert pe._settle_frames() > 0
    r.stop()
np.save(sys.argv[1], np.concatenate(outs))
''')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for exact in ("0", "1"):
        out = tmp_path / f"y{exact}.npy"
        env = dict(os.environ, PGX_ROOT=root, PGX_LADDER_EXACT_EMIT=exact)
        p = subprocess.run([sys.executable, str(script), str(out)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
        got[exact] = np.load(out)
    peak = float(np.max(np.abs(got["1"])))
    err = float(np.max(np.abs(got["0"].astype(np.float64) - got["1"])))
    assert err <= 2e-7 * peak, (err, peak)
    assert np.mean(got["0"] != got["1"]) < 0.05


# ---------------------------------------------------------------------------- at and above self-oscillation
def _resonant(spec_source, res, look, optimistic, blocks, freq=800.0, drive=1.5):
    import pygmu2_amd as pg
    import spec_build
    from pygmu2_amd import device, ladder_pe, look_ahead
    from oracle.golden_cases import S
    pg.set_sample_rate(44100)
    spec = S("LadderPE", source=spec_source, frequency=freq, resonance=res, mode="lp24", drive=drive, oversample=2)
    keep = ladder_pe.OPTIMISTIC
    ladder_pe.OPTIMISTIC = optimistic
    look_ahead.set_enabled(look)
    try:
        pe = spec_build.build(spec)
        r = pg.NullRenderer(sample_rate=44100)
        r.set_source(pe)
        r.start()
        outs = []
        for s, n in blocks:
            outs.append(pe.render(s, n).data.copy())
            if pe._optimist is not None:
                pe._optimist.poll(wait=True)             # deterministic here: the verdict before the next render
        state = pe._state.to_host().copy()
        opt = pe._optimist
        r.stop()
        return outs, state, opt, spec
    finally:
        ladder_pe.OPTIMISTIC = keep
        look_ahead.set_enabled(True)


@pytest.mark.parametrize("res", [0.6, 0.9, 1.0])
def test_driven_ladder_above_self_oscillation_runs_in_time_segments(res):
    """The reference's own example setting (examples/17_ladder_filter.py:43: 800 Hz, resonance 0.6, drive 1.5) and
    stronger ones: ladder_settle_frames has no answer (the small-signal loop oscillates), the warm-up length is found by
    trial, the device check accepts it -- a saw locks the saturating loop to itself -- and the samples are the
    sequential kernel's and the oracle's."""
    from oracle import graph_eval
    from oracle.golden_cases import S
    from pygmu2_amd.ladder_pe import ladder_settle_frames, STATS
    assert ladder_settle_frames(800.0, res, 44100.0, 2) == 0
    saw = S("BlitSawPE", frequency=110.0)
    blocks = [(i * 44100, 44100) for i in range(6)]
    before = dict(STATS)
    got, st, opt, spec = _resonant(saw, res, False, True, blocks)
    assert opt is not None and STATS["segmented"] - before["segmented"] >= 5
    assert opt.sleep == 0 and opt.good >= 2, "a length that the device check accepts was found"
    want_seq, st_seq, _, _ = _resonant(saw, res, False, False, blocks)
    g = graph_eval.Node(spec, 44100)
    peak = 0.0
    for (s, n), a, b in zip(blocks, got, want_seq):
        o = g.render(s, n)
        peak = max(peak, float(np.max(np.abs(o))))
        assert float(np.max(np.abs(a.astype(np.float64) - b))) <= 1e-6 * peak
        assert float(np.max(np.abs(a.astype(np.float64) - o))) <= REL_TOL * peak
    assert np.allclose(st, st_seq, rtol=1e-6, atol=1e-8)


def test_free_running_ladder_gives_up_and_stays_exact():
    """A lone sine beside a ladder at resonance 0.9 does not entrain it (the loop oscillates at its own pitch and
    phase): every trial length fails the device check, what disagrees is repaired sequentially on the device -- from
    the first bad boundary on, in the reference's operation order -- and the PE then stays on the sequential kernel.
    (The stream's first segments, where the ladder is still at rest beside a tiny input, pass the check and keep the
    fused arithmetic of a segment's own samples: the same trajectory to ~1e-10, not the same bits.)"""
    from oracle.golden_cases import S
    from pygmu2_amd.ladder_pe import OPTIMISTIC_SETTLES
    sine = S("SinePE", frequency=220.0, amplitude=0.5)
    blocks = [(i * 100_000, 100_000) for i in range(len(OPTIMISTIC_SETTLES) + 2)]
    got, st, opt, _ = _resonant(sine, 0.9, False, True, blocks)
    assert opt.sleep > 0 and opt.seen == len(OPTIMISTIC_SETTLES), (opt.sleep, opt.seen, opt.level)
    want, st_seq, _, _ = _resonant(sine, 0.9, False, False, blocks)
    peak = max(float(np.max(np.abs(b))) for b in want)
    for a, b in zip(got, want):
        assert float(np.max(np.abs(a.astype(np.float64) - b))) <= 1e-6 * peak
    assert np.allclose(st, st_seq, rtol=1e-6, atol=1e-9)


def test_one_bad_boundary_is_repaired_where_it_is(env):
    """k_ladder_finish repairs what disagrees, not the chain: a stable ladder rendered with a warm-up far too short for
    ONE stretch of the block (a loud burst after silence: the zero-started segments there are off) -- the counters say
    how many samples were rendered again, the output is the sequential kernel's to the segmented path's bound."""
    n = 200_000
    x = _signal(n, 1, 9)
    x[:150_000] *= 1e-3                                           # quiet, then loud: the loud part forgets slower (drive)
    settle = 96                                                   # (the estimate for this setting is ~1 000)
    y_seq, st_seq, _, _ = _ladder(env, x, 0, res=0.45, freq=900.0)
    device, lib = env.device, env.lib
    xin = device.DeviceBuffer.from_host(x)
    out = device.DeviceBuffer((n, 1), np.float32)
    params = device.upload_struct(device.LADDER_PARAMS, freq=900.0, resonance=0.45, drive=1.0, passband_gain=0.5,
                                  oversample=2, mode=0)
    st = device.DeviceBuffer.from_host(np.zeros((1, 9)))
    need = lib.pgx_ladder_workspace_bytes(1, n, 1, settle)
    ws = device.DeviceBuffer((need,), np.uint8, zero=True)
    device.check(lib.pgx_ladder(out.ptr, 0, xin.ptr, 0, 1, n, 1, 48000.0, params.ptr, None, None, None, st.ptr, settle,
                                0, ws.ptr))
    head = ws.to_host()[:16]
    chains, launches, repaired = int(head.view(np.int32)[0]), int(head.view(np.int32)[1]), int(head.view(np.int64)[1])
    assert chains == 1 and launches == 1 and 0 < repaired <= n, (chains, launches, repaired)
    y = out.to_host()
    peak = float(np.max(np.abs(y_seq)))
    assert float(np.max(np.abs(y.astype(np.float64) - y_seq))) <= 1e-6 * peak
    assert np.allclose(st.to_host(), st_seq, rtol=1e-6, atol=1e-9)


def test_resonant_ladder_through_look_ahead_windows():
    """The same driven ladder streamed in 4096-frame blocks: windows of 8, 16, 32 ... blocks, each one segmented
    launch; against block-by-block sequential rendering."""
    from oracle.golden_cases import S
    ssaw = S("SuperSawPE", frequency=110.0, voices=7, seed=3)
    blocks = [(i * 4096, 4096) for i in range(70)]
    got, _, opt, _ = _resonant(ssaw, 0.6, True, True, blocks)
    want, _, _, _ = _resonant(ssaw, 0.6, False, False, blocks)
    peak = max(float(np.max(np.abs(b))) for b in want)
    for i, (a, b) in enumerate(zip(got, want)):
        assert float(np.max(np.abs(a.astype(np.float64) - b))) <= 2e-6 * peak, i
