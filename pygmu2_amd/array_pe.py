"""ArrayPE: a finite source backed by a host array, resident in HBM after first use
(array_pe.py:45-129)."""

from __future__ import annotations

import numpy as np

from ._kernels import DeviceBuffer, check, lib, new_output
from .extent import ExtendMode, Extent
from .snippet import Snippet
from .source_pe import SourcePE


class ArrayPE(SourcePE):
    _READ_AHEAD_SAFE = True

    def __init__(self, data, extend_mode: ExtendMode = ExtendMode.ZERO):
        arr = np.asarray(data, dtype=np.float32)
        if arr.ndim == 1:
            arr = arr.reshape(-1, 1)
        elif arr.ndim > 2:
            raise ValueError(f"ArrayPE data must be 1D or 2D, got {arr.ndim}D")
        if arr.shape[0] == 0:
            raise ValueError("ArrayPE data cannot be empty")
        self._data = arr
        self._length, self._channels = arr.shape
        self._extend_mode = extend_mode
        self._dev: DeviceBuffer | None = None

    @property
    def data(self) -> np.ndarray:
        return self._data

    def channel_count(self) -> int:
        return self._channels

    def _compute_extent(self) -> Extent:
        return Extent(0, self._length)

    def _render(self, start: int, duration: int) -> Snippet:
        if self._dev is None:
            self._dev = DeviceBuffer.from_host(np.ascontiguousarray(self._data))
        if start >= 0 and start + duration <= self._length:
            # entirely inside the array: hand out the resident rows themselves (Snippets are read-only)
            return Snippet.window_rows(start, self._dev, start, duration)
        out = new_output(duration, self._channels)
        hold_first = self._extend_mode in (ExtendMode.HOLD_FIRST, ExtendMode.HOLD_BOTH)
        hold_last = self._extend_mode in (ExtendMode.HOLD_LAST, ExtendMode.HOLD_BOTH)
        check(lib().pgx_window_copy(out.ptr, start, duration, self._channels, self._dev.ptr, 0,
                                    self._length, int(hold_first), int(hold_last)), "pgx_window_copy")
        return Snippet(start, out)

    def __repr__(self) -> str:
        ext = f", extend_mode={self._extend_mode.value}" if self._extend_mode != ExtendMode.ZERO else ""
        return f"ArrayPE(shape={self._data.shape}{ext})"
