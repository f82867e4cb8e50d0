"""CPU: host-side contract of the PE layer (no device calls).

Mirrors the behaviours the reference pins in tests/test_extent.py, tests/test_snippet.py,
tests/test_processing_element.py:203-241 and tests/test_renderer.py:194-542, using
hand-rolled host PEs so that nothing touches the GPU."""

import numpy as np
import pytest

import pygmu2_amd as pg
from pygmu2_amd import Extent, ExtendMode, NullRenderer, ProcessingElement, Snippet, SourcePE


# ------------------------------------------------------------------ Extent
def test_extent_basics_and_infinite_bounds():
    e = Extent(10, 20)
    assert (e.start, e.end, e.duration) == (10, 20, 10)
    assert e.contains(10) and e.contains(19) and not e.contains(20) and not e.contains(9)
    assert Extent(None, 5).contains(-10 ** 9) and not Extent(None, 5).contains(5)
    assert Extent(None, None).duration is None
    with pytest.raises(ValueError):
        Extent(5, 4)


def test_extent_empty_is_falsy_and_never_intersects():
    z = Extent(7, 7)
    assert z.is_empty() and not z and bool(Extent(0, 1))
    assert not z.intersects(Extent(None, None)) and not Extent(0, 10).intersects(z)
    assert Extent(0, 10).union(z) == Extent(0, 10) and z.union(Extent(0, 10)) == Extent(0, 10)


def test_extent_intersection_union_spans():
    a, b = Extent(0, 10), Extent(5, 15)
    assert a.intersects(b) and a.intersection(b) == Extent(5, 10) and a.union(b) == Extent(0, 15)
    assert not Extent(0, 5).intersects(Extent(5, 9))
    d = Extent(0, 5).intersection(Extent(8, 9))
    assert d.is_empty() and d.start == 8
    assert Extent(None, 10).intersection(Extent(3, None)) == Extent(3, 10)
    assert Extent(None, 10).union(Extent(3, 12)) == Extent(None, 12)
    assert a.spans(2, 8) and not a.spans(2, 9) and a.spans(100, 0)
    # the idiom the filters rely on: empty intersection falls back
    assert (Extent(0, 5).intersection(Extent(9, 12)) or Extent(0, 5)) == Extent(0, 5)


# ------------------------------------------------------------------ Snippet (host payloads)
def test_snippet_normalises_shape_and_dtype():
    s = Snippet(5, np.arange(4, dtype=np.float64))
    assert s.data.shape == (4, 1) and s.data.dtype == np.float32
    assert (s.start, s.end, s.duration, s.channels) == (5, 9, 4, 1)
    assert not s.on_device
    z = Snippet.from_zeros(-3, 0, 2)
    assert z.duration == 0 and z.channels == 2 and z.end == -3
    with pytest.raises(ValueError):
        Snippet(0, np.zeros((2, 2, 2)))
    assert Snippet(0, np.ones((3, 2))) == Snippet(0, np.ones((3, 2), np.float32))
    assert Snippet(0, np.ones((3, 2))) != Snippet(1, np.ones((3, 2)))


# ------------------------------------------------------------------ mock PEs (host only)
class HostRamp(SourcePE):
    def __init__(self, channels=1, pure=True):
        self._ch, self._pure = channels, pure
        self.events = []

    def channel_count(self):
        return self._ch

    def is_pure(self):
        return self._pure

    def _render(self, start, duration):
        idx = np.arange(start, start + duration, dtype=np.float32).reshape(-1, 1)
        return Snippet(start, np.tile(idx, (1, self._ch)))

    def _on_start(self):
        self.events.append("start")

    def _on_stop(self):
        self.events.append("stop")


class HostPass(ProcessingElement):
    def __init__(self, *srcs, pure=True, need=None, log=None, name=""):
        self._srcs, self._pure, self._need, self._log, self._name = list(srcs), pure, need, log, name

    def inputs(self):
        return self._srcs

    def is_pure(self):
        return self._pure

    def required_input_channels(self):
        return self._need

    def _render(self, start, duration):
        return self._srcs[0].render(start, duration)

    def _on_start(self):
        if self._log is not None:
            self._log.append(("start", self._name))

    def _on_stop(self):
        if self._log is not None:
            self._log.append(("stop", self._name))


def test_sample_rate_required_before_construction():
    pg.config._sample_rate = None
    with pytest.raises(RuntimeError):
        HostRamp()
    pg.set_sample_rate(48000)
    assert HostRamp().sample_rate == 48000


def test_render_contract_negative_zero_and_dispatch():
    pe = HostRamp(channels=2)
    with pytest.raises(ValueError):
        pe.render(0, -1)
    z = pe.render(17, 0)
    assert z.duration == 0 and z.channels == 2 and z.start == 17
    s = pe.render(-2, 4)
    assert s.data[:, 0].tolist() == [-2, -1, 0, 1]
    assert pe.extent() == Extent(None, None)


def test_scalar_or_pe_values_helper():
    pe = HostPass(HostRamp())
    v = pe._scalar_or_pe_values(0.5, 0, 4)
    assert v.dtype == np.float64 and v.shape == (4,) and np.all(v == 0.5)
    v = pe._scalar_or_pe_values(HostRamp(channels=2), 3, 4)
    assert v.tolist() == [3, 4, 5, 6]
    v = pe._scalar_or_pe_values(HostRamp(channels=2), 3, 4, allow_multichannel=True, dtype=np.float32)
    assert v.shape == (4, 2) and v.dtype == np.float32
    assert pe._scalar_or_pe_values(1.0, 0, 0).shape == (0,)
    assert pe._scalar_or_pe_values(2, 0, 3, allow_multichannel=True, channels=2).shape == (3, 2)
    with pytest.raises(ValueError):
        pe._scalar_or_pe_values(HostRamp(), 0, 4, channel=1)


# ------------------------------------------------------------------ Renderer
def test_renderer_rejects_shared_stateful_pe_and_channel_mismatch():
    shared = HostRamp(pure=False)
    root = HostPass(HostPass(shared), HostPass(shared))
    with pytest.raises(ValueError, match="multiple sinks"):
        NullRenderer().set_source(root)
    NullRenderer().set_source(HostPass(HostPass(HostRamp()), HostPass(HostRamp())))
    pure_shared = HostRamp(pure=True)
    NullRenderer().set_source(HostPass(HostPass(pure_shared), HostPass(pure_shared)))
    with pytest.raises(ValueError, match="requires 1 channel"):
        NullRenderer().set_source(HostPass(HostRamp(channels=2), need=1))
    r = NullRenderer(sample_rate=48000)
    r.set_source(HostPass(HostRamp(channels=2)))
    assert r.channel_count == 2 and r.sample_rate == 48000


def test_renderer_lifecycle_order_and_errors():
    log = []
    leaf = HostPass(HostRamp(), log=log, name="leaf")
    mid = HostPass(leaf, log=log, name="mid")
    root = HostPass(mid, leaf, log=log, name="root")          # diamond: leaf visited once
    r = NullRenderer()
    with pytest.raises(RuntimeError):
        r.start()
    r.set_source(root)
    with pytest.raises(RuntimeError):
        r.render(0, 16)
    r.start()
    assert log == [("start", "leaf"), ("start", "mid"), ("start", "root")]
    with pytest.raises(RuntimeError):
        r.start()
    with pytest.raises(RuntimeError):
        r.set_source(root)
    with pytest.raises(ValueError):
        r.render(0, 0)
    r.render(0, 16)
    assert r.last_snippet.duration == 16
    log.clear()
    r.stop()
    assert log == [("stop", "root"), ("stop", "mid"), ("stop", "leaf")]
    r.stop()                                                   # idempotent
    assert log == [("stop", "root"), ("stop", "mid"), ("stop", "leaf")]
    with NullRenderer() as ctx:
        ctx.set_source(HostRamp())
        ctx.start()
    assert not ctx.started


def test_renderer_lenient_mode_warns_instead_of_raising():
    pg.set_error_mode(pg.ErrorMode.LENIENT)
    try:
        r = NullRenderer()
        r.set_source(HostRamp())
        r.start()
        r.start()                      # warning only
        with pytest.raises(RuntimeError):
            NullRenderer().start()     # fatal regardless of mode
    finally:
        pg.set_error_mode(pg.ErrorMode.STRICT)


def test_renderer_profiling_report():
    r = NullRenderer(sample_rate=1000)
    r.set_source(HostRamp())
    r.enable_profiling()
    r.start()
    r.render(0, 100)
    r.render(100, 100)
    rep = r.get_profile_report()
    assert rep.render_calls == 2 and rep.total_samples == 200 and rep.total_render_time_ns > 0
    (prof,) = rep.pe_profiles.values()
    assert prof.pe_class == "HostRamp" and prof.render_count == 2 and prof.total_samples == 200
    assert prof.min_time_ns <= prof.max_time_ns and prof.samples_per_second > 0 and prof.realtime_ratio(1000) > 0
    text = rep.summary(1000)
    assert "RENDER PROFILE REPORT" in text and "Total render calls: 2" in text and "HostRamp" in text


# ------------------------------------------------------------------ PE graph plumbing that needs no device
def test_pe_static_properties_without_device():
    pg.set_sample_rate(48000)
    sine = pg.SinePE(440.0, channels=2)
    assert sine.is_pure() and sine.channel_count() == 2 and sine.inputs() == []
    fm = pg.SinePE(frequency=pg.SinePE(5.0))
    assert not fm.is_pure() and len(fm.inputs()) == 1
    with pytest.raises(ValueError):
        pg.MixPE(sine)
    mix = pg.MixPE([pg.CropPE(sine, 0, 100), pg.CropPE(sine, 50, 100)])
    assert mix.extent() == Extent(0, 150) and mix.channel_count() == 2
    with pytest.raises(ValueError, match="channel mismatch"):
        mix.resolve_channel_count([1, 2])
    assert mix.resolve_channel_count([2, 2]) == 2
    arr = pg.ArrayPE([1.0, 2.0, 3.0])
    assert arr.extent() == Extent(0, 3) and arr.channel_count() == 1
    with pytest.raises(ValueError):
        pg.ArrayPE([])
    bq = pg.BiquadPE(arr, frequency=pg.CropPE(pg.ConstantPE(500.0), 10, 5), q=0.7)
    assert bq.extent() == Extent(0, 3)                 # empty intersection falls back
    lad = pg.LadderPE(arr, frequency=pg.CropPE(pg.ConstantPE(500.0), 10, 5))
    assert lad.extent().is_empty()                     # strict intersection
    conv = pg.ConvolvePE(pg.ArrayPE(np.zeros(10)), pg.ArrayPE([1.0, 0.5, 0.25]))
    assert conv.extent() == Extent(0, 12) and not conv.is_pure()
    with pytest.raises(ValueError):
        pg.ConvolvePE(arr, pg.ConstantPE(1.0)).extent()
    with pytest.raises(ValueError):
        pg.ConvolvePE(arr, pg.CropPE(pg.ArrayPE([1, 0, 0]), 1, 2)).extent()
    with pytest.raises(ValueError):
        pg.CropPE(sine, 0, -1)
    with pytest.raises(ValueError):
        pg.PeriodicTrigger(0.0)
    assert pg.PeriodicTrigger(7.0, phase=0.25)._period == 6857
    with pytest.raises(ValueError):
        pg.SuperSawPE(440.0, mix_mode="bogus")


def test_supersaw_tables_match_reference_values():
    """tests/test_super_saw_pe.py:221-327 style pins (+ the SURVEY KAT phases)."""
    pg.set_sample_rate(44100)
    s = pg.SuperSawPE(440.0, voices=7, seed=1234)
    assert np.allclose([float(o.initial_phase[0]) for o in s._oscillators],
                       [0.97669977, 0.38019574, 0.92324623, 0.26169242, 0.31909706, 0.11809123,
                        0.24176629], atol=1e-8)
    assert np.isclose(np.sum(np.asarray(s._mix_gains, dtype=np.float64) ** 2), 1.0, atol=1e-6)
    assert np.isclose(s._detune_ratios[3], 1.0) and np.isclose(s._detune_ratios[0], 2 ** (-20 / 1200))
    lin = pg.SuperSawPE(440.0, voices=5, mix_mode="linear")._mix_gains
    assert np.allclose(lin / lin[2], [0.5, 0.75, 1.0, 0.75, 0.5])
    ch = pg.SuperSawPE(440.0, voices=6, mix_mode="center_heavy")._mix_gains
    assert np.allclose(ch / ch[2], [0.5, 0.5, 1, 1, 0.5, 0.5])
    assert len(pg.SuperSawPE(440.0, voices=0)._oscillators) == 1
    assert pg.SuperSawPE(440.0).inputs() == []          # internal oscillators are hidden


def test_gate_validation_on_host_arrays():
    pg.GateSignal._validate_gate_array(np.array([[0.0], [1.0]], np.float32))
    with pytest.raises(ValueError):
        pg.GateSignal._validate_gate_array(np.array([[0.5]], np.float32))
    with pytest.raises(ValueError):
        pg.GateSignal._validate_gate_array(np.zeros((4, 2), np.float32))
    pg.TriggerSignal._validate_trigger_array(np.array([[0.0], [2.0], [-1.0]], np.float32))
    with pytest.raises(ValueError):
        pg.TriggerSignal._validate_trigger_array(np.array([[0.25]], np.float32))


# ------------------------------------------------------------------ SVF / envelope / transform host logic

def test_svf_rejects_allpass_and_lists_inputs():
    src = pg.ConstantPE(0.5)
    with pytest.raises(ValueError, match="ALLPASS"):
        pg.SVFilterPE(src, 1000.0, 0.7, mode=pg.BiquadMode.ALLPASS)
    f = pg.SinePE(frequency=2.0, amplitude=100.0)
    pe = pg.SVFilterPE(src, f, 0.7)
    assert pe.inputs() == [src, f] and not pe.is_pure() and pe.channel_count() == 1
    assert "SVFilterPE" in repr(pe) and "SinePE(...)" in repr(pe)


def test_svf_host_coefficients_match_oracle():
    from oracle import pe_oracle
    from pygmu2_amd.svfilter_pe import svf_coefficients
    for mode in pe_oracle.SVF_MODES:
        for f, q, g in ((1500.0, 1.3, 4.5), (30000.0, 0.001, -9.0), (0.01, 500.0, 12.0)):
            A, B, Cc = pe_oracle.svf_coefficients_batch([f], [q], mode, g, 44100)
            want = (A[0, 0, 0], A[0, 0, 1], A[0, 1, 0], A[0, 1, 1], B[0, 0], B[0, 1], Cc[0, 0], Cc[0, 1], Cc[0, 2])
            got = svf_coefficients(pg.BiquadMode(mode), f, q, g, 44100)
            assert got == tuple(float(v) for v in want), (mode, f, q, g)


def test_envelope_clamps_parameters():
    src = pg.ConstantPE(0.5)
    pe = pg.EnvelopePE(src, attack=-1.0, release=0.2, lookahead=5.0)
    assert pe.attack == 0.0 and pe.lookahead == 0.0 and pe.release == 0.2
    pe = pg.EnvelopePE(src, attack=0.01, release=0.2, lookahead=5.0, mode=pg.DetectionMode.RMS)
    assert pe.lookahead == 0.01 and pe.mode is pg.DetectionMode.RMS and not pe.is_pure()
    assert pe.extent().start is None


def test_transform_descriptors_are_numpy_callables():
    from pygmu2_amd import transforms as tf
    from oracle import pe_oracle
    x = np.linspace(-2, 2, 41, dtype=np.float32).reshape(-1, 1)
    spec = [["clip", 0.0, 1.0], ["sqrt"], ["affine", 2900.0, 100.0], ["one_minus"], ["square"], ["abs"], ["tanh"]]
    chain = tf.from_spec(spec)
    assert [op[0] for op in chain.ops()] == [tf.CLIP, tf.SQRT, tf.AFFINE, tf.ONE_MINUS, tf.SQUARE, tf.ABS, tf.TANH]
    want = pe_oracle.transform(x, spec)
    assert np.array_equal(chain(x.astype(np.float64)).astype(np.float32), want)
    assert isinstance(tf.lower(np.abs), tf.Abs) and isinstance(tf.lower(np.tanh), tf.Tanh)
    assert tf.lower(lambda v: v * 2) is None
    with pytest.raises(ValueError):
        tf.Clip(1.0, 0.0)
    with pytest.raises(TypeError):
        tf.Chain(np.abs)
    pe = pg.TransformPE(pg.ConstantPE(1.0), func=tf.Affine(2.0, 1.0))
    assert pe.is_pure() and pe.name == "affine" and "func=affine" in repr(pe)
    assert pg.TransformPE(pg.ConstantPE(1.0), func=np.tanh, name="soft").name == "soft"


def test_ladder_settle_estimate_follows_the_small_signal_loop():
    from pygmu2_amd.ladder_pe import ladder_settle_frames as settle
    base = settle(1200.0, 0.3, 48000, 2)
    assert 500 < base < 3000 and base % 32 == 0
    assert settle(1200.0, 0.0, 48000, 2) < base < settle(1200.0, 0.5, 48000, 2)   # resonance slows forgetting
    assert settle(200.0, 0.3, 48000, 2) > base > settle(6000.0, 0.3, 48000, 2)    # so does a low cutoff
    assert settle(1200.0, 0.6, 48000, 2) == 0          # k*q_adjust > 4: the linear loop oscillates
    assert settle(2000.0, 1.0, 48000, 2) == 0
    assert settle(20.0, 0.3, 48000, 2) == 0            # forgets too slowly to be worth segmenting
    pe = pg.LadderPE(pg.ConstantPE(0.1), frequency=pg.SinePE(frequency=1.0, amplitude=100.0), resonance=0.3)
    assert pe._settle_frames() == 0                    # PE-driven cutoff: always the sequential kernel


def test_biquad_settle_frames_bounds_the_state_matrix():
    from pygmu2_amd.biquad_pe import rbj_coefficients, settle_frames
    c = rbj_coefficients(pg.BiquadMode.LOWPASS, 1000.0, 0.707, 0.0, 44100.0)
    w = settle_frames(c[3], c[4])
    assert w == 1024
    a = np.array([[-c[3], 1.0], [-c[4], 0.0]])
    assert np.max(np.abs(np.linalg.matrix_power(a, w))) < 2.0 ** -90
    assert np.max(np.abs(np.linalg.matrix_power(a, w // 2))) >= 2.0 ** -90
    assert settle_frames(-1.999, 0.9991) == 0          # poles at radius ~0.9995: no usable horizon
    assert settle_frames(-2.1, 1.2) == 0               # unstable


def test_kemar_grid_and_nearest_file_rule():
    from pygmu2_amd.spatial_pe import SpatialHRTF, kemar_entries
    entries = kemar_entries()
    assert len(entries) == 368 and entries[0] == (-40, 0, "H-40e000a.wav") and entries[-1] == (90, 0, "H90e000a.wav")
    assert (50, 176, "H50e176a.wav") in entries and (-40, 6, "H-40e006a.wav") in entries
    assert SpatialHRTF.hrtf_filename_for(45.0, 0.0) == "H0e045a.wav"
    assert SpatialHRTF.hrtf_filename_for(-30.0, 2.0) == "H0e030a.wav"          # mirrored hemisphere, nearest elevation
    assert SpatialHRTF.hrtf_filename_for(200.0, 95.0) == "H80e180a.wav"          # azimuth saturates at 180
    assert SpatialHRTF.hrtf_filename_for(3.0, 95.0) == "H90e000a.wav"
    with pytest.raises(ValueError, match="static"):
        SpatialHRTF(pg.SinePE(frequency=1.0), 0.0)
    with pytest.raises(ValueError, match="channels must be >= 1"):
        pg.SpatialAdapter(0)
    pe = pg.SpatialPE(pg.ConstantPE(0.5), method=pg.SpatialConstantPower(pg.SinePE(frequency=0.5, amplitude=90.0)))
    assert pe.channel_count() == 2 and pe.is_pure() and len(pe.inputs()) == 2
    assert "SpatialConstantPower(azimuth=SinePE)" in repr(pe)


# ------------------------------------------------------------------------------------------- round 3 host logic
def test_fusing_gate_of_the_sine_biquad_chain():
    """BiquadPE(SinePE) is rendered as one launch only where the filter passes more of the tone than of the float32
    rounding of its samples (biquad_pe._passes_more_signal_than_rounding)."""
    from pygmu2_amd.biquad_pe import BiquadMode, _passes_more_signal_than_rounding as gate, rbj_coefficients
    sr = 44100.0

    def coef(mode, fc, q):
        return tuple(float(x) for x in np.ravel(rbj_coefficients(BiquadMode(mode), fc, q, 0.0, sr)))

    w = lambda f: 2.0 * np.pi * f / sr
    assert gate(coef("lowpass", 1000.0, 0.707), w(440.0), 1024)          # BASELINE config 2
    assert gate(coef("bandpass", 2500.0, 2.0), w(3000.3), 1024)
    assert gate(coef("highpass", 3000.0, 0.707), w(5500.0), 1024)
    assert not gate(coef("highpass", 8000.0, 0.707), w(55.0), 1024)      # passes the rounding noise, not the tone
    assert not gate(coef("lowpass", 200.0, 0.707), w(15000.0), 1024)
    assert not gate((float("nan"), 0.0, 0.0, 0.0, 0.0), w(440.0), 1024)


def test_comb_scalar_delay_follows_the_reference_expression():
    """comb_pe.py:61-77 with the smoother settled on a scalar frequency: D = clip(round(sr / max(max(f, min_f), 1)), 1,
    buffer_len - 1), buffer_len = ceil(sr / min_f) + 1 (comb_pe.py:216-218); numpy's round is half-to-even."""
    pg.set_sample_rate(44100)
    src = pg.ConstantPE(0.25, channels=2)
    assert pg.CombPE(src, 440.0, 0.7)._scalar_delay() == 100
    assert pg.CombPE(src, 440.0, 0.7)._buffer_rows() == 2206
    assert pg.CombPE(src, 5.0, 0.7)._scalar_delay() == 2205                      # clamped to min_frequency = 20 Hz
    assert pg.CombPE(src, 1e9, 0.7)._scalar_delay() == 1
    pg.set_sample_rate(48000)
    assert pg.CombPE(src, 30000.0, 0.99)._scalar_delay() == 2                    # 1.6 -> 2
    assert pg.CombPE(src, 19200.0, 0.5)._scalar_delay() == 2                     # 2.5 -> 2 (half to even)
    assert pg.CombPE(src, 13714.285714285714, 0.5)._scalar_delay() in (3, 4)     # 3.5 +- an ulp
    rec = pg.CombPE(src, 220.0, 0.9, min_frequency=30.0)._param_record()
    assert int(rec["delay"][0]) == 218 and int(rec["buffer_len"][0]) == 1601 and float(rec["feedback"][0]) == 0.9
    pg.set_sample_rate(44100)


def test_look_ahead_windows_double():
    from pygmu2_amd import look_ahead
    assert look_ahead.FIRST_WINDOW_BLOCKS == 8 and look_ahead.WINDOW_GROWTH == 2 and look_ahead.AHEAD_BLOCKS == 256
    sizes, grow = [], look_ahead.FIRST_WINDOW_BLOCKS
    for _ in range(7):
        sizes.append(min(grow, look_ahead.AHEAD_BLOCKS))
        grow *= look_ahead.WINDOW_GROWTH
    assert sizes == [8, 16, 32, 64, 128, 256, 256]
    # a 20-block stream (the driver's bench run): one plain block, then windows of 8 and 16 -> 25 rendered, not 42
    assert 1 + sizes[0] + sizes[1] == 25


def test_comb_and_small_oscillator_banks_are_batchable():
    from pygmu2_amd import voice_bank
    pg.set_sample_rate(48000)
    comb = lambda i: pg.CombPE(pg.BlitSawPE(frequency=110.0 + i), frequency=220.0 + i, feedback=0.5)
    assert voice_bank._signature(comb(0)) == ("comb", ("blitsaw", 1))
    assert voice_bank._signature(pg.CombPE(pg.BlitSawPE(110.0), frequency=pg.SinePE(2.0), feedback=0.5)) is None
    assert voice_bank._signature(pg.CombPE(pg.BlitSawPE(110.0), frequency=220.0, feedback=pg.SinePE(2.0))) is None
    pg.set_sample_rate(44100)


def _raw_library():
    """The C-ABI library without a device: planning entry points only (nothing that launches)."""
    import ctypes
    from pygmu2_amd import build
    lib = ctypes.CDLL(build.LIB_PATH)
    lib.pgx_supersaw_wide_segments.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64]
    lib.pgx_supersaw_wide_table_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.pgx_supersaw_wide_table_bytes.restype = ctypes.c_size_t
    lib.pgx_convolve_fft_size.argtypes = [ctypes.c_int64]
    lib.pgx_convolve_fft_size.restype = ctypes.c_int64
    lib.pgx_convolve_fft_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64]
    lib.pgx_convolve_fft_workspace_bytes.restype = ctypes.c_size_t
    return lib


def test_segment_planner_of_the_wide_bank():
    """pgx_supersaw_wide_segments: the plan with the smallest estimated makespan (host arithmetic only)."""
    lib = _raw_library()
    seg = lib.pgx_supersaw_wide_segments
    assert seg(64, 7, 48_000) == 4            # 12 tiles: 4 segments x 3 tiles = one workgroup per CU
    assert seg(128, 7, 48_000) == 4           # two workgroups per CU beat one with six tiles
    assert seg(512, 7, 48_000) == 1           # enough instances: no segment entry cost
    assert seg(64, 7, 4096) == 1              # a single tile
    assert seg(1, 7, 2_822_400) == 230        # a lone SuperSaw over a 64-block window: 689 tiles in threes -- the chip is full
    assert seg(0, 7, 48_000) == 1 and seg(64, 0, 48_000) == 1 and seg(64, 7, 0) == 1
    for batch in (1, 3, 17, 64, 200, 512, 5000):
        for nv in (1, 7, 16):
            for n in (1, 4095, 4097, 48_000, 1 << 20):
                k = seg(batch, nv, n)
                tiles = -(-n // 4096)
                assert 1 <= k <= tiles
                assert (k - 1) * -(-tiles // k) < tiles          # no empty segment
    assert lib.pgx_supersaw_wide_table_bytes(3, 7) == 3 * 7 * 216 * 8


def test_fft_convolution_sizes():
    lib = _raw_library()
    assert lib.pgx_convolve_fft_size(65536) == 131072 and lib.pgx_convolve_fft_size(2048) == 4096
    assert lib.pgx_convolve_fft_size(131072) == 262144 and lib.pgx_convolve_fft_size(131073) == 0
    # 96 000 stereo frames, 65 536 taps: two overlap-save blocks per channel -> two packed transforms of 2^17 complex doubles
    need = lib.pgx_convolve_fft_workspace_bytes(96_000, 65536, 2, 131072)
    assert need >= 2 * 131072 * 16 + 65535 * 2 * 4


def test_look_ahead_frame_cap_grows_only_for_small_graphs(monkeypatch):
    """A window root with one or two PEs under it takes four times the frames (one or two buffers of that size, not dozens);
    PGX_LOOK_AHEAD_FRAMES pins the cap for every graph."""
    from pygmu2_amd import look_ahead
    monkeypatch.setattr(look_ahead, "_FRAMES_EXPLICIT", False)
    base, small = look_ahead.AHEAD_FRAMES, look_ahead.AHEAD_FRAMES_SMALL_GRAPH
    assert look_ahead.frame_cap(1) == look_ahead.frame_cap(2) == max(base, small)
    assert look_ahead.frame_cap(3) == look_ahead.frame_cap(4) == max(base, small // 2)
    assert look_ahead.frame_cap(5) == look_ahead.frame_cap(500) == base
    monkeypatch.setattr(look_ahead, "_FRAMES_EXPLICIT", True)
    assert look_ahead.frame_cap(1) == base


def test_settle_frames_fine_is_the_first_multiple_of_16_that_settles():
    """The on-chip mix's warm-up: the smallest multiple of 16 frames W with every entry of A^W below 2^-90 -- never above
    settle_frames' power of two, 0 where the section does not settle within the limit."""
    import numpy as np
    from pygmu2_amd.biquad_pe import BiquadMode, rbj_coefficients, settle_frames, settle_frames_fine
    for f, q in ((2000.0, 0.707), (500.0, 2.0), (8000.0, 0.5), (120.0, 0.9)):
        c = rbj_coefficients(BiquadMode.LOWPASS, f, q, 0.0, 48000)
        w, w2 = settle_frames_fine(c[3], c[4]), settle_frames(c[3], c[4])
        if w:
            a = np.array([[-c[3], 1.0], [-c[4], 0.0]])
            assert w % 16 == 0 and 0 < w <= 2048 and (not w2 or w <= w2)
            assert np.all(np.abs(np.linalg.matrix_power(a, w)) < 2.0 ** -90)
            assert not np.all(np.abs(np.linalg.matrix_power(a, w - 16)) < 2.0 ** -90)
    ringing = rbj_coefficients(BiquadMode.LOWPASS, 500.0, 400.0, 0.0, 48000)
    assert settle_frames_fine(ringing[3], ringing[4]) == 0
