"""DiracPE: unit impulse at sample index 0 (dirac_pe.py:45-67)."""

from __future__ import annotations

from ._kernels import check, lib, new_output
from .snippet import Snippet
from .source_pe import SourcePE


class DiracPE(SourcePE):
    _READ_AHEAD_SAFE = True

    def __init__(self, channels: int = 1):
        self._channels = channels

    def _render(self, start: int, duration: int) -> Snippet:
        out = new_output(duration, self._channels)
        check(lib().pgx_dirac(out.ptr, start, duration, self._channels), "pgx_dirac")
        return Snippet(start, out)

    def channel_count(self) -> int:
        return self._channels

    def __repr__(self) -> str:
        return f"DiracPE(channels={self._channels})"
