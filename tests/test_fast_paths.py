"""CPU: the C hot paths (pygmu2_amd/csrc/_fast.c) behave like the Python code they replace -- ProcessingElement.render's
two window exits, the hand-over of everything else to the Python implementation, Snippet.__del__'s hooks."""

import gc

import numpy as np
import pytest

import pygmu2_amd as pg
from pygmu2_amd import look_ahead, processing_element as P
from pygmu2_amd.snippet import Snippet


class _Window:            # stands in for a resident (frames, channels) DeviceBuffer: the row view only asks for .shape
    def __init__(self, frames, channels):
        self.shape = (frames, channels)


class _Source(pg.SourcePE):
    def __init__(self):
        self.calls = []

    def channel_count(self):
        return 2

    def _render(self, start, duration):
        self.calls.append((start, duration))
        return pg.Snippet(start, np.zeros((duration, 2), dtype=np.float32))


pytestmark = pytest.mark.skipif(not P.FAST_PATHS, reason="pygmu2_amd/_fast.so not built")


def test_read_ahead_window_exit():
    pg.set_sample_rate(48000)
    pe = _Source()
    win = _Window(64 * 1024, 2)
    pe.__dict__["_ra_win"] = (1000, 1000 + 64 * 1024, win)
    s = pe.render(1000 + 3 * 1024, 1024)
    assert not pe.calls, "served from the window: nothing rendered"
    assert (s.start, s.duration, s.channels) == (1000 + 3 * 1024, 1024, 2)
    assert s._base == (win, 3 * 1024) and s._base[0] is win and s._bank_window is False
    assert s._ready is None and s._copy is None and s._host is None and s._dev is None
    assert pe.__dict__["_ra_last"] == 1000 + 4 * 1024 and s.on_device
    pe.__dict__.pop("_ra_win")
    pe.__dict__["_ra_ok"] = False
    out = pe.render(5, 7)                                     # outside any window: the Python path, which renders
    assert pe.calls == [(5, 7)] and out.duration == 7


def test_look_ahead_window_exit_and_its_conditions():
    pg.set_sample_rate(48000)
    pe = _Source()
    pe.__dict__["_la_ok"] = False                             # (nothing below may open a window of its own)
    win = look_ahead._Window()
    win.first, win.end, win.buf, win.served, win.snap, win.nodes, win.block = 0, 8 * 256, _Window(8 * 256, 2), 256, [], [], 0
    pe.__dict__["_la_win"] = win
    s = pe.render(256, 256)
    assert not pe.calls and s._base == (win.buf, 256) and win.served == 512 and pe.__dict__["_la_last"] == 512
    s2 = pe.render(512, 100)                                  # any length while the window is not block-sensitive
    assert s2.duration == 100 and win.served == 612
    win.block = 256
    win.served = 768
    s3 = pe.render(768, 256)
    assert s3._base == (win.buf, 768) and win.served == 1024
    # a pull that does not continue the stream settles the window (Python path): the window goes, the PE renders
    pe.render(0, 64)
    assert "_la_win" not in pe.__dict__ and pe.calls[-1] == (0, 64)


def test_everything_else_is_the_python_implementation():
    pg.set_sample_rate(48000)
    pe = _Source()
    pe.__dict__["_ra_ok"] = pe.__dict__["_la_ok"] = False
    with pytest.raises(ValueError):
        pe.render(0, -1)
    z = pe.render(10, 0)
    assert z.duration == 0 and z.channels == 2 and not pe.calls
    assert pe.render(start=3, duration=4).duration == 4       # keywords
    assert pe.render(np.int64(7), np.int64(2)).start == 7     # not exact ints: handed over, same result
    with pytest.raises(TypeError):
        pe.render(1)


def test_snippet_finalizer_hooks_still_run():
    fired = []

    def ready():
        fired.append("ready")
    s = Snippet(0, np.zeros((4, 1), dtype=np.float32), ready=ready)
    del s
    gc.collect()
    assert fired == ["ready"]

    def ready2():
        fired.append("ready2")
    ready2.on_drop = lambda: fired.append("dropped")
    s = Snippet(0, np.zeros((4, 1), dtype=np.float32), ready=ready2)
    del s
    gc.collect()
    assert fired == ["ready", "dropped"]
    s = Snippet.window_rows(0, _Window(16, 1), 0, 4)          # nothing pending: the C exit
    del s


def test_read_ahead_window_cut_for_one_block_length():
    """(first, end, buffer, period): only the blocks of the window's grid are served from it."""
    pg.set_sample_rate(48000)
    pe = _Source()
    pe.__dict__["_ra_ok"] = pe.__dict__["_la_ok"] = False
    win = _Window(64 * 100, 2)
    pe.__dict__["_ra_win"] = (1000, 1000 + 64 * 100, win, 100)
    s = pe.render(1300, 100)
    assert not pe.calls and s._base == (win, 300)
    pe.render(1350, 100)                                      # off the grid
    pe.render(1400, 50)                                       # another length
    assert pe.calls == [(1350, 100), (1400, 50)]
