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


def _first_known(*candidates):
    for c in candidates:
        if c is not None:
            return c
    return None


class LoopPE(ProcessingElement):
    """Behaviour: frames [region_first, region_last) of the source repeat from output time 0 -- for ever, or
    `count` times and then silence.  A boundary the caller leaves open falls back to the source's extent
    (the start further to 0; an open end over an endless source is an error).  An optional crossfade, given
    in seconds and never longer than half the region, blends the end of each pass into its beginning."""

    def __init__(self, source: ProcessingElement, loop_start: int | None = None, loop_end: int | None = None,
                 count: int | None = None, crossfade_seconds: float | None = None):
        if crossfade_seconds is not None and crossfade_seconds < 0:
            raise ValueError(f"crossfade_seconds must be non-negative, got {crossfade_seconds}")
        self._source = source
        self._loop_start, self._loop_end = loop_start, loop_end
        self._count = count
        self._crossfade_seconds = crossfade_seconds
        self._loop_snippet: Snippet | None = None         # kept only under a pure sub-graph
        self._sample_rate = source.sample_rate            # the loop lives at its source's rate

        span = source.extent()
        first = _first_known(loop_start, span.start, 0)
        last = _first_known(loop_end, span.end)
        if last is None:
            raise ValueError("Cannot loop source with infinite extent without explicit loop_end")
        if last - first <= 0:
            raise ValueError(f"Loop length must be positive, got {last - first}")
        self._region = (first, last)
        self._loop_length = last - first

        wanted = 0
        if crossfade_seconds is not None and self._sample_rate is not None:
            wanted = int(round(crossfade_seconds * self.sample_rate))
        self._crossfade = min(wanted, self._loop_length // 2)

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

    # every output frame is a gather by its absolute index into a region that, under a pure source, is rendered
    # once: small sequential pulls are served from read-ahead windows
    _READ_AHEAD_SAFE = True

    def _read_ahead_condition(self) -> bool:
        return _subtree_pure(self._source)

    def channel_count(self) -> int | None:
        return self._source.channel_count()

    def _played_frames(self) -> int | None:
        return None if self._count is None else self._count * self._loop_length

    def _compute_extent(self) -> Extent:
        return Extent(0, self._played_frames())

    def _drop_region(self) -> None:
        self._loop_snippet = None

    _on_start = _on_stop = _drop_region

    def _render(self, start: int, duration: int) -> Snippet:
        played = self._played_frames()
        if played is not None and start >= played:
            # past the last pass: silence, and the source is not pulled at all
            return Snippet(start, new_output(duration, self._source.channel_count() or 1, zero=True))
        region = self._loop_snippet
        if region is None:
            region = self._source.render(self._region[0], self._loop_length)
            if _subtree_pure(self._source):
                self._loop_snippet = region               # a pure source renders the same region every time
        out = new_output(duration, region.channels)
        check(lib().pgx_loop(out.ptr, region.dev.ptr, start, duration, region.channels, self._loop_length,
                             -1 if played is None else played, self._crossfade), "pgx_loop")
        return Snippet(start, out)

    def __repr__(self) -> str:
        parts = [f"source={type(self._source).__name__}", f"loop_start={self._loop_start}",
                 f"loop_end={self._loop_end}"]
        if self._count is not None:
            parts.append(f"count={self._count}")
        if self._crossfade_seconds:
            parts.append(f"crossfade_seconds={self._crossfade_seconds}")
        return "LoopPE(" + ", ".join(parts) + ")"
