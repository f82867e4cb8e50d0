"""
CombPE: feedback comb y[n] = x[n] + fb * y[n - D], D = round(sr / smoothed_freq)
(comb_pe.py:124-349).  With a scalar frequency the delay is one integer: the comb is D independent
first-order recurrences that run lane-parallel (and, on long renders, in concurrent time segments);
the float64 ring only carries the last outputs from one render to the next.  With a frequency PE the
delays come from a time-parallel evaluation of the reference's one-pole and the ring runs in LDS
(csrc/pgx_comb.hip).
"""

from __future__ import annotations

import numpy as np

from . import device as _dev
from ._kernels import DeviceBuffer, check, lib, new_output, ptr
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


class CombPE(ProcessingElement):
    _LOOK_AHEAD_SAFE = True            # look_ahead.py
    _STATE_FIELDS = ("_ring", "_state", "_buffer_len", "_total", "_parity")

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
        self._ring: DeviceBuffer | None = None       # (2, buffer_len, C) float64: the half `_parity` is current
        self._state: DeviceBuffer | None = None      # {smoothed_freq, entry level, tie flag, blocks redone} (frequency PE only)
        self._buffer_len = 0
        self._total = 0                              # frames rendered since the reset: write_pos = total % buffer_len
        self._parity = 0
        self._params: DeviceBuffer | None = None
        self._ws: DeviceBuffer | None = None

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

    def _buffer_rows(self) -> int:
        max_delay = int(np.ceil(self.sample_rate / self._min_frequency))      # comb_pe.py:216-218
        return max(2, max_delay + 1)

    def _scalar_delay(self) -> int:
        """comb_pe.py:61-77 for a scalar frequency: the smoother equals its input from the first sample on."""
        raw = max(float(self._frequency), self._min_frequency)
        d = int(np.rint(float(self.sample_rate) / max(raw, 1.0)))             # np.round: half to even
        return min(max(d, 1), self._buffer_rows() - 1)

    def _param_record(self) -> np.ndarray:
        rec = np.zeros(1, dtype=_dev.COMB_PARAMS)
        rec[0] = (0.0 if self._fb_is_pe else float(self._feedback),
                  0 if self._freq_is_pe else self._scalar_delay(), self._buffer_rows())
        return rec

    def _allocate(self, channels: int) -> None:
        self._buffer_len = self._buffer_rows()
        self._ring = DeviceBuffer((2, self._buffer_len, channels), np.float64, zero=True)
        self._state = DeviceBuffer.from_host(np.array([-1.0, 0.0, 0.0, 0.0], dtype=np.float64))
        self._total = 0
        self._parity = 0
        if self._params is None:
            self._params = _dev.upload_structs(self._param_record())

    def _on_start(self) -> None:
        self._allocate(self._source.channel_count() or 1)

    def _on_stop(self) -> None:
        self._ring = None
        self._state = None

    def _render(self, start: int, duration: int) -> Snippet:
        src = self._source.render(start, duration)
        ch = src.channels
        if self._ring is None or self._ring.shape[2] != ch:
            self._allocate(ch)
        f_s, f_buf = self._control_stream(self._frequency, start, duration)
        b_s, b_buf = self._control_stream(self._feedback, start, duration)
        L = lib()
        delay = 0 if self._freq_is_pe else self._scalar_delay()
        # (with a frequency stream the size depends on the ring's rows, not on a delay)
        need = L.pgx_comb_workspace_bytes(1, duration, ch, self._buffer_len if self._freq_is_pe else delay,
                                          1 if self._freq_is_pe else 0)
        if need and (self._ws is None or self._ws.nbytes < need):
            self._ws = DeviceBuffer((need,), np.uint8)
        out = new_output(duration, ch)
        check(L.pgx_comb(out.ptr, 0, src.dev.ptr, 0, 1, duration, ch, float(self.sample_rate), self._params.ptr,
                         delay, delay, ptr(f_buf), ptr(b_buf), self._min_frequency, self._smoothing_samples,
                         self._ring.ptr, self._buffer_len, self._total, self._parity, self._state.ptr,
                         ptr(self._ws) if need else None),
              "pgx_comb")
        self._total += duration
        self._parity ^= 1
        return Snippet(start, out)

    def __repr__(self) -> str:
        f = f"{type(self._frequency).__name__}(...)" if self._freq_is_pe else self._frequency
        b = f"{type(self._feedback).__name__}(...)" if self._fb_is_pe else self._feedback
        return f"CombPE(source={type(self._source).__name__}, frequency={f}, feedback={b})"
