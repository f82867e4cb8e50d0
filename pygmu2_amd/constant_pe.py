"""ConstantPE: a constant-valued source of infinite extent (constant_pe.py:42-63)."""

from __future__ import annotations

from ._kernels import check, lib, new_output
from .snippet import Snippet
from .source_pe import SourcePE


class ConstantPE(SourcePE):
    _READ_AHEAD_SAFE = True

    def __init__(self, value: float, channels: int = 1):
        self._value = value
        self._channels = channels

    @property
    def value(self) -> float:
        return self._value

    def _render(self, start: int, duration: int) -> Snippet:
        out = new_output(duration, self._channels)
        # np.full(..., value, dtype=float32): the value is rounded to float32 once
        check(lib().pgx_fill(out.ptr, duration * self._channels, float(self._value)), "pgx_fill")
        return Snippet(start, out)

    def channel_count(self) -> int:
        return self._channels

    def __repr__(self) -> str:
        return f"ConstantPE(value={self._value}, channels={self._channels})"
