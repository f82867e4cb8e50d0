"""GainPE: multiply a source by a scalar or by a gain PE (gain_pe.py:59-155).  One
float32 multiply per sample, so the result is bit-identical to the reference's."""

from __future__ import annotations

import numpy as np

from ._kernels import check, lib, new_output
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


class GainPE(ProcessingElement):
    _PASSES_BLOCKS = True              # look_ahead.py: inputs are pulled with the caller's (duration)
    _READ_AHEAD_SAFE = True

    def __init__(self, source: ProcessingElement, gain=1.0):
        self._source = source
        self._gain = gain
        self._gain_is_pe = isinstance(gain, ProcessingElement)

    source = property(lambda self: self._source)
    gain = property(lambda self: self._gain)

    def inputs(self) -> list[ProcessingElement]:
        return [self._source, self._gain] if self._gain_is_pe else [self._source]

    def is_pure(self) -> bool:
        return True

    def channel_count(self) -> int | None:
        return self._source.channel_count()

    def _compute_extent(self) -> Extent:
        ext = self._source.extent()
        return ext.intersection(self._gain.extent()) if self._gain_is_pe else ext

    def _render(self, start: int, duration: int) -> Snippet:
        if not self._gain_is_pe and duration > 0:
            fused = getattr(self._source, "_render_with_gain", None)
            if fused is not None and self._source.is_pure():
                # producer + constant gain in one launch (same float32 roundings as two launches)
                return fused(start, duration, float(np.float32(self._gain)))
        src = self._source.render(start, duration)
        if not self._gain_is_pe and self._gain == 1.0:
            return src                                           # x * float32(1) is x, bit for bit
        ch = src.channels
        out = new_output(duration, ch)
        if self._gain_is_pe:
            g = self._gain.render(start, duration)
            gch = g.channels
            if gch != 1 and gch != ch:
                # numpy would refuse to broadcast (N, gch) against (N, ch)
                raise ValueError(f"operands could not be broadcast together with shapes "
                                 f"({duration},{ch}) ({duration},{gch})")
            check(lib().pgx_gain_vec(out.ptr, src.dev.ptr, g.dev.ptr, duration, ch, gch), "pgx_gain_vec")
        else:
            # the reference multiplies by np.float32(gain): round the scalar to float32 first
            g32 = self.__dict__.get("_gain_f32")
            if g32 is None:
                g32 = self.__dict__["_gain_f32"] = float(np.float32(self._gain))
            check(lib().pgx_gain_const(out.ptr, src.dev.ptr, duration * ch, g32), "pgx_gain_const")
        return Snippet(start, out)

    def __repr__(self) -> str:
        g = f"{type(self._gain).__name__}(...)" if self._gain_is_pe else str(self._gain)
        return f"GainPE(source={type(self._source).__name__}, gain={g})"
