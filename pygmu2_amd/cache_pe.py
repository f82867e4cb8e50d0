"""CachePE: one-entry memo of the last (start, duration) render (cache_pe.py:34-81).
The cached Snippet keeps its device payload alive, so a repeated pull costs nothing."""

from __future__ import annotations

from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


class CachePE(ProcessingElement):
    _PASSES_BLOCKS = True              # look_ahead.py: inputs are pulled with the caller's (duration)
    _LOOK_AHEAD_SAFE = True            # look_ahead.py: the memo is restored with the other states ...
    _STATE_FIELDS = ("_key", "_snippet")

    def _look_ahead_condition(self) -> bool:
        # ... but only over a pure source is the memo invisible: a stateful source pulled twice per block (an
        # EnvelopePE with look-ahead next to the dry path) depends on how the stream is cut into blocks
        from . import read_ahead
        return read_ahead.eligible(self._source)

    def __init__(self, source: ProcessingElement):
        self._source = source
        self._key: tuple[int, int] | None = None
        self._snippet: Snippet | None = None

    source = property(lambda self: self._source)

    def inputs(self) -> list[ProcessingElement]:
        return [self._source]

    def is_pure(self) -> bool:
        return True          # declared pure by the reference (cache_pe.py:47-50)

    def channel_count(self) -> int | None:
        return self._source.channel_count()

    def _compute_extent(self) -> Extent:
        return self._source.extent()

    def _reset_state(self) -> None:
        self._key = None
        self._snippet = None

    _on_start = _reset_state
    _on_stop = _reset_state

    def _render(self, start: int, duration: int) -> Snippet:
        if self._snippet is not None and self._key == (start, duration):
            return self._snippet
        self._snippet = self._source.render(start, duration)
        self._key = (start, duration)
        return self._snippet

    def __repr__(self) -> str:
        return f"CachePE(source={type(self._source).__name__})"
