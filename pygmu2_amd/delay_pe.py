"""
DelayPE: integer, fractional or PE-driven delay (delay_pe.py:13-240).

Integer delays re-address the source render (no data is touched).  Fractional and PE
delays render the source window the indices reach and interpolate on the device
(pgx_interp_lookup: linear or Catmull-Rom, interpolated_lookup.py:28-130) with the
reference's expression order, so the output is the reference's bit for bit.
For a PE delay the window bounds are data dependent: the device reduces min/max of the
indices (pgx_index_range) and the host reads those 16 bytes back.
"""

from __future__ import annotations

from enum import Enum

import numpy as np

from ._kernels import DeviceBuffer, check, lib, new_output
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


class InterpolationMode(Enum):
    LINEAR = "linear"
    CUBIC = "cubic"


class DelayPE(ProcessingElement):
    def __init__(self, source: ProcessingElement, delay, interpolation: InterpolationMode = InterpolationMode.LINEAR):
        self._source = source
        self._delay = delay
        self._interpolation = interpolation
        if isinstance(delay, ProcessingElement):
            self._mode = "pe"
        elif isinstance(delay, float) and not delay.is_integer():
            self._mode = "float"
        else:
            self._mode = "int"
            self._delay = int(delay)
        self._range_dev: DeviceBuffer | None = None

    source = property(lambda self: self._source)
    delay = property(lambda self: self._delay)
    interpolation = property(lambda self: self._interpolation)

    def inputs(self) -> list[ProcessingElement]:
        return [self._source, self._delay] if self._mode == "pe" else [self._source]

    def is_pure(self) -> bool:
        return True

    def channel_count(self) -> int | None:
        return self._source.channel_count()

    def _compute_extent(self) -> Extent:
        if self._mode == "pe":
            return self._source.extent().intersection(self._delay.extent())
        ext = self._source.extent()
        amount = self._delay
        new_start = None if ext.start is None else ext.start + amount
        new_end = None if ext.end is None else ext.end + amount
        if self._mode == "float":
            new_start = None if new_start is None else int(np.floor(new_start))
            new_end = None if new_end is None else int(np.ceil(new_end))
        return Extent(new_start, new_end)

    def _render(self, start: int, duration: int) -> Snippet:
        if self._mode == "int":
            snip = self._source.render(start - self._delay, duration)
            return Snippet(start, snip.dev)
        cubic = getattr(self._interpolation, "value", self._interpolation) == "cubic"
        if self._mode == "float":
            delay_buf = None
            # t - delay is monotonic: its extrema are the end points (same float64 operations)
            idx_min = float(np.float64(start) - self._delay)
            idx_max = float(np.float64(start + duration - 1) - self._delay)
        else:
            _, delay_buf = self._control_stream(self._delay, start, duration)
            if self._range_dev is None:
                self._range_dev = DeviceBuffer((2,), np.float64)
            check(lib().pgx_index_range(self._range_dev.ptr, delay_buf.ptr, start, duration), "pgx_index_range")
            idx_min, idx_max = (float(v) for v in self._range_dev.to_host())
            if not (np.isfinite(idx_min) and np.isfinite(idx_max)):
                raise ValueError("DelayPE: delay stream contains non-finite values")
        margin = 2 if cubic else 1
        needed_min = int(np.floor(idx_min)) - (margin - 1)
        needed_len = int(np.ceil(idx_max)) + margin - needed_min
        window = self._source.render(needed_min, needed_len)
        ch = window.channels
        ext = self._source.extent()
        bounded = ext.start is not None and ext.end is not None
        out = new_output(duration, ch)
        check(lib().pgx_interp_lookup(out.ptr, window.dev.ptr, needed_min, needed_len, ch, start, duration,
                                      0.0 if delay_buf is not None else float(self._delay),
                                      None if delay_buf is None else delay_buf.ptr, int(cubic), int(bounded),
                                      float(ext.start) if bounded else 0.0, float(ext.end) if bounded else 0.0),
              "pgx_interp_lookup")
        return Snippet(start, out)

    def __repr__(self) -> str:
        src = type(self._source).__name__
        if self._mode == "pe":
            return (f"DelayPE(source={src}, delay={type(self._delay).__name__}(...), "
                    f"interpolation={self._interpolation.value})")
        if self._mode == "float":
            return f"DelayPE(source={src}, delay={self._delay}, interpolation={self._interpolation.value})"
        return f"DelayPE(source={src}, delay={self._delay})"
