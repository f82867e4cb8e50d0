"""NullRenderer: pulls the graph and discards the result (null_renderer.py:30-32); the
driver used by the benchmark harness.  `sync_output=True` makes every render wait for the
device, which is what a consumer of the samples would observe."""

from __future__ import annotations

from . import device as _dev
from .renderer import Renderer
from .snippet import Snippet


class NullRenderer(Renderer):
    def __init__(self, sample_rate: int = 44100, sync_output: bool = False):
        super().__init__(sample_rate)
        self._sync_output = sync_output
        self.last_snippet: Snippet | None = None

    def _output(self, snippet: Snippet) -> None:
        self.last_snippet = snippet
        if self._sync_output and snippet.on_device:
            _dev.synchronize()
