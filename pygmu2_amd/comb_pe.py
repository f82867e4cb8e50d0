"""
CombPE: feedback comb y[n] = x[n] + fb * y[n - D], D = round(sr / smoothed_freq)
(comb_pe.py:124-349).  The delay line is a float64 ring buffer in HBM; the integer delay
sequence comes from the reference's exact one-pole smoothing recurrence, then samples
that do not reach into their own chunk are processed in parallel (pgx_comb).
"""

from __future__ import annotations

import numpy as np

from ._kernels import DeviceBuffer, check, lib, new_output, ptr
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


class CombPE(ProcessingElement):
    _LOOK_AHEAD_SAFE = True            # look_ahead.py
    _STATE_FIELDS = ("_ring", "_state", "_buffer_len")

    _MAX_FEEDBACK = 0.995

    def __init__(self, source: ProcessingElement, frequency, feedback=0.0,
                 min_frequency: float = 20.0, smoothing_samples: int = 2400):
        self._source = source
        self._frequency = frequency
        self._feedback = feedback
        self._min_frequency = max(1.0, float(min_frequency))
        self._smoothing_samples = max(1, int(smoothing_samples))
        self._freq_is_pe = isinstance(frequency, ProcessingElement)
        self._fb_is_pe = isinstance(feedback, ProcessingElement)
        self._ring: DeviceBuffer | None = None       # (buffer_len, C) float64
        self._state: DeviceBuffer | None = None      # {write_pos, smoothed_freq}
        self._buffer_len = 0

    source = property(lambda self: self._source)
    frequency = property(lambda self: self._frequency)
    feedback = property(lambda self: self._feedback)

    def inputs(self) -> list[ProcessingElement]:
        out = [self._source]
        if self._freq_is_pe:
            out.append(self._frequency)
        if self._fb_is_pe:
            out.append(self._feedback)
        return out

    def is_pure(self) -> bool:
        return False

    def channel_count(self) -> int | None:
        return self._source.channel_count()

    def _compute_extent(self) -> Extent:
        ext = self._source.extent()
        if self._freq_is_pe:
            ext = ext.intersection(self._frequency.extent()) or ext
        if self._fb_is_pe:
            ext = ext.intersection(self._feedback.extent()) or ext
        return ext

    def _allocate(self, channels: int) -> None:
        max_delay = int(np.ceil(self.sample_rate / self._min_frequency))
        self._buffer_len = max(2, max_delay + 1)
        self._ring = DeviceBuffer((self._buffer_len, channels), np.float64, zero=True)
        self._state = DeviceBuffer.from_host(np.array([0.0, -1.0], dtype=np.float64))

    def _on_start(self) -> None:
        self._allocate(self._source.channel_count() or 1)

    def _on_stop(self) -> None:
        self._ring = None
        self._state = None

    def _render(self, start: int, duration: int) -> Snippet:
        src = self._source.render(start, duration)
        ch = src.channels
        if self._ring is None or self._ring.shape[1] != ch:
            self._allocate(ch)
        f_s, f_buf = self._control_stream(self._frequency, start, duration)
        b_s, b_buf = self._control_stream(self._feedback, start, duration)
        delay = DeviceBuffer((duration,), np.int32)
        fbv = DeviceBuffer((duration,), np.float64)
        out = new_output(duration, ch)
        check(lib().pgx_comb(out.ptr, src.dev.ptr, duration, ch, float(self.sample_rate),
                             0.0 if f_s is None else f_s, 0.0 if b_s is None else b_s,
                             ptr(f_buf), ptr(b_buf), self._min_frequency, self._smoothing_samples,
                             self._ring.ptr, self._buffer_len, self._state.ptr, delay.ptr, fbv.ptr),
              "pgx_comb")
        return Snippet(start, out)

    def __repr__(self) -> str:
        f = f"{type(self._frequency).__name__}(...)" if self._freq_is_pe else self._frequency
        b = f"{type(self._feedback).__name__}(...)" if self._fb_is_pe else self._feedback
        return f"CombPE(source={type(self._source).__name__}, frequency={f}, feedback={b})"
