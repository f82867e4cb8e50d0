"""
WindowPE: centred (zero-phase) sliding max / min / mean / RMS of the source (window_pe.py:24-258).

The source is rendered with half a window of padding on both sides and reduced on the device (pgx_window):
64-frame block statistics first, then head + whole blocks + tail per output frame.  max / min are exact; mean
and RMS sum each window directly in float64 where the reference differences a cumulative sum.
"""

from __future__ import annotations

from enum import Enum

import numpy as np

from ._kernels import DeviceBuffer, check, lib, new_output
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


class WindowMode(Enum):
    MAX = "max"
    MEAN = "mean"
    RMS = "rms"
    MIN = "min"


_MODE_CODE = {WindowMode.MAX: 0, WindowMode.MIN: 1, WindowMode.MEAN: 2, WindowMode.RMS: 3}


class WindowPE(ProcessingElement):
    def __init__(self, source: ProcessingElement, window: float = 0.05, mode: WindowMode = WindowMode.MAX,
                 rectify: bool = True):
        self._source = source
        self._window = max(0.0, window)
        self._mode = mode
        self._rectify = rectify
        self._workspace: DeviceBuffer | None = None

    source = property(lambda self: self._source)
    window = property(lambda self: self._window)
    mode = property(lambda self: self._mode)
    rectify = property(lambda self: self._rectify)

    def inputs(self) -> list[ProcessingElement]:
        return [self._source]

    def is_pure(self) -> bool:
        return True

    # max / min of a window do not depend on how the stream is cut into blocks (the float64 sums of mean / RMS are
    # grouped by 64-frame blocks counted from the request's start: equal only to rounding): read-ahead for those --
    # over a PURE source only.  The padded pulls of two neighbouring blocks overlap, so a stateful PE below is pulled
    # out of sequence and starts over at every block (window_pe.py:135-141 does the same): what a block holds then
    # depends on where it begins, and a window rendered in one piece is not the blocks it is cut into.
    _READ_AHEAD_SAFE = True

    def _read_ahead_condition(self) -> bool:
        from .loop_pe import _subtree_pure
        return self._mode in (WindowMode.MAX, WindowMode.MIN) and _subtree_pure(self._source)

    def channel_count(self) -> int | None:
        return self._source.channel_count()

    def _compute_extent(self) -> Extent:
        return self._source.extent()

    def _render(self, start: int, duration: int) -> Snippet:
        half = max(1, int(self._window * self.sample_rate / 2))                 # window_pe.py:135-136
        src = self._source.render(start - half, duration + 2 * half)
        ch = src.channels
        L = lib()
        need = L.pgx_window_workspace_bytes(duration, ch, half)
        if self._workspace is None or self._workspace.nbytes < need:
            self._workspace = DeviceBuffer((need,), np.uint8)
        out = new_output(duration, ch)
        check(L.pgx_window(out.ptr, src.dev.ptr, duration, ch, half, _MODE_CODE[self._mode],
                           1 if self._rectify else 0, self._workspace.ptr), "pgx_window")
        return Snippet(start, out)

    def __repr__(self) -> str:
        return (f"WindowPE(source={self._source.__class__.__name__}, window={self._window}, "
                f"mode={self._mode.value}, rectify={self._rectify})")
