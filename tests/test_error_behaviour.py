"""CPU: the error behaviour the reference's own tests pin for the path's classes (exception type and the
message fragment they match), checked against this package.  No device needed: every check fires in host
logic before any kernel would run.

reference tests mirrored: test_mix_pe.py:30-37,226; test_convolve_pe.py:29-41; test_extent.py:44-49;
test_snippet.py:46-50; test_renderer.py:158-178,232-244,378-435."""

import numpy as np
import pytest

import pygmu2_amd as pg
from pygmu2_amd import Extent, NullRenderer, Snippet


@pytest.fixture(autouse=True)
def _rate():
    pg.set_sample_rate(44100)
    pg.set_error_mode(pg.ErrorMode.STRICT)


def test_mix_needs_two_inputs_and_matching_channels():
    with pytest.raises(ValueError, match="at least 2 inputs"):
        pg.MixPE(pg.ConstantPE(0.5))
    with pytest.raises(ValueError, match="at least 2 inputs"):
        pg.MixPE()
    mix = pg.MixPE(pg.ConstantPE(0.5, channels=1), pg.ConstantPE(0.5, channels=2))
    with pytest.raises(ValueError, match="channel mismatch"):
        mix.resolve_channel_count([1, 2])


def test_convolve_filter_extent_rules():
    src = pg.ArrayPE([1, 2, 3, 4])
    with pytest.raises(ValueError):
        pg.ConvolvePE(src, pg.CropPE(pg.ArrayPE([1, 0, 0]), 1, 2)).extent()         # filter must start at 0
    with pytest.raises(ValueError):
        pg.ConvolvePE(src, pg.ConstantPE(1.0)).extent()                              # ... and be finite
    assert pg.ConvolvePE(src, pg.ArrayPE([1.0, 0.5, -1.0])).extent() == Extent(0, 6)


def test_extent_and_snippet_constructors():
    with pytest.raises(ValueError):
        Extent(200, 100)
    assert Extent(100, 100).is_empty() is True
    with pytest.raises(ValueError):
        Snippet(0, np.zeros((2, 2, 2)))
    assert Snippet(5, np.zeros(0)).duration == 0
    one_d = Snippet(0, np.arange(4))
    assert one_d.data.shape == (4, 1) and one_d.data.dtype == np.float32


class _StereoOnly(pg.ProcessingElement):
    def __init__(self, source):
        self._source = source

    def inputs(self):
        return [self._source]

    def required_input_channels(self):
        return 2

    def channel_count(self):
        return 2

    def _render(self, start, duration):
        return self._source.render(start, duration)


class _Stateful(pg.ProcessingElement):
    def __init__(self, source):
        self._source = source

    def inputs(self):
        return [self._source]

    def is_pure(self):
        return False

    def channel_count(self):
        return self._source.channel_count()

    def _render(self, start, duration):
        return self._source.render(start, duration)


def test_renderer_lifecycle_errors():
    r = NullRenderer(sample_rate=44100)
    with pytest.raises(RuntimeError, match="No source set"):
        r.render(0, 10)
    with pytest.raises(RuntimeError, match="No source set"):
        r.start()
    r.set_source(pg.ConstantPE(1.0))
    with pytest.raises(RuntimeError, match="Not started"):
        r.render(0, 10)
    r.start()
    with pytest.raises(RuntimeError, match="Already started"):
        r.start()
    with pytest.raises(RuntimeError, match="Cannot set source while started"):
        r.set_source(pg.ConstantPE(2.0))
    with pytest.raises(ValueError, match="duration >= 1"):
        r.render(0, 0)
    r.stop()
    r.stop()                                                    # stopping twice is fine


def test_renderer_graph_validation():
    r = NullRenderer(sample_rate=44100)
    shared = _Stateful(pg.ConstantPE(1.0))
    with pytest.raises(ValueError, match="not pure"):
        r.set_source(pg.MixPE(shared, shared))
    with pytest.raises(ValueError, match="requires 2 channel"):
        r.set_source(_StereoOnly(pg.ConstantPE(1.0, channels=1)))
    r.set_source(_StereoOnly(pg.ConstantPE(1.0, channels=2)))
    assert r.channel_count == 2


def test_render_argument_checks_do_not_need_a_device():
    pe = pg.SinePE(440.0)
    with pytest.raises(ValueError):
        pe.render(0, -1)
    empty = pe.render(123, 0)
    assert empty.duration == 0 and empty.start == 123 and empty.channels == 1
