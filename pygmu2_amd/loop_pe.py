"""
LoopPE: repeat a segment of the source (loop_pe.py:17-252).

The loop region is rendered once per render call (as the reference does) -- or once for good when the whole
sub-graph under it is pure -- and every output frame is a modular gather into it (pgx_loop); the optional
crossfade blends the end of the loop into its beginning with the reference's float64 weights.  Bit-exact.
"""

from __future__ import annotations

from ._kernels import check, lib, new_output
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


def _subtree_pure(pe: ProcessingElement) -> bool:
    return pe.is_pure() and all(_subtree_pure(i) for i in pe.inputs())


class LoopPE(ProcessingElement):
    def __init__(self, source: ProcessingElement, loop_start: int | None = None, loop_end: int | None = None,
                 count: int | None = None, crossfade_seconds: float | None = None):
        if crossfade_seconds is not None and crossfade_seconds < 0:
            raise ValueError(f"crossfade_seconds must be non-negative, got {crossfade_seconds}")
        self._source = source
        self._loop_start = loop_start
        self._loop_end = loop_end
        self._count = count
        self._crossfade_seconds = crossfade_seconds
        self._resolved_start: int | None = None
        self._resolved_end: int | None = None
        self._loop_length: int | None = None
        self._crossfade = 0
        self._loop_snippet: Snippet | None = None         # kept only under a pure sub-graph
        self._sample_rate = self._source.sample_rate        # loop_pe.py:64
        self._resolve_loop_boundaries()
        if self._sample_rate is not None:
            self._resolve_crossfade()

    source = property(lambda self: self._source)
    loop_start = property(lambda self: self._loop_start)
    loop_end = property(lambda self: self._loop_end)
    count = property(lambda self: self._count)
    crossfade_seconds = property(lambda self: float(self._crossfade_seconds or 0.0))
    crossfade_samples = property(lambda self: int(self._crossfade))

    def inputs(self) -> list[ProcessingElement]:
        return [self._source]

    def is_pure(self) -> bool:
        return True

    def channel_count(self) -> int | None:
        return self._source.channel_count()

    def _compute_extent(self) -> Extent:
        if self._loop_length is None or self._count is None:
            return Extent(0, None)
        return Extent(0, self._count * self._loop_length)

    def _resolve_loop_boundaries(self) -> None:      # loop_pe.py:122-148
        ext = self._source.extent()
        if self._resolved_start is None:
            if self._loop_start is not None:
                self._resolved_start = self._loop_start
            elif ext.start is not None:
                self._resolved_start = ext.start
            else:
                self._resolved_start = 0
        if self._resolved_end is None:
            if self._loop_end is not None:
                self._resolved_end = self._loop_end
            elif ext.end is not None:
                self._resolved_end = ext.end
            else:
                raise ValueError("Cannot loop source with infinite extent without explicit loop_end")
        self._loop_length = self._resolved_end - self._resolved_start
        if self._loop_length <= 0:
            raise ValueError(f"Loop length must be positive, got {self._loop_length}")

    def _resolve_crossfade(self) -> None:             # loop_pe.py:150-157
        if self._crossfade_seconds is not None:
            self._crossfade = int(round(self._crossfade_seconds * self.sample_rate))
        else:
            self._crossfade = 0
        if self._loop_length is not None:
            self._crossfade = min(self._crossfade, self._loop_length // 2)

    def _on_start(self) -> None:
        self._loop_snippet = None

    _on_stop = _on_start

    def _render(self, start: int, duration: int) -> Snippet:
        channels = self._source.channel_count() or 1
        total = -1 if self._count is None else self._count * self._loop_length
        if total >= 0 and (start >= total or min(duration, total - start) <= 0):
            return Snippet(start, new_output(duration, channels, zero=True))      # no source pull (loop_pe.py:176-187)
        loop = self._loop_snippet
        if loop is None:
            loop = self._source.render(self._resolved_start, self._loop_length)
            if _subtree_pure(self._source):
                self._loop_snippet = loop
        out = new_output(duration, loop.channels)
        check(lib().pgx_loop(out.ptr, loop.dev.ptr, start, duration, loop.channels, self._loop_length, total,
                             self._crossfade), "pgx_loop")
        return Snippet(start, out)

    def __repr__(self) -> str:
        count_str = f", count={self._count}" if self._count is not None else ""
        xfade_str = f", crossfade_seconds={self._crossfade_seconds}" if self._crossfade_seconds else ""
        return (f"LoopPE(source={self._source.__class__.__name__}, loop_start={self._loop_start}, "
                f"loop_end={self._loop_end}{count_str}{xfade_str})")
