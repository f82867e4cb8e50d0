"""
TriggerRestartPE: restart a source at local time 0 on every trigger event
(trigger_restart_pe.py:13-98).  Pure control flow: the trigger block is read back to find
the event positions (4 bytes per frame), each stretch between events is one render of the
source placed into the output with a device-to-device copy.
"""

from __future__ import annotations

import numpy as np

from ._kernels import check, lib, new_output
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet
from .trigger_signal import TriggerSignal


class TriggerRestartPE(ProcessingElement):
    """Behaviour: every positive trigger sample re-bases the source's clock -- the source is reset and then
    rendered from its own time 0 at that frame; until the next trigger it simply keeps running on that
    clock, across block boundaries.  Before the first trigger ever seen the output is silence.  Channels and
    state follow the source, the extent follows the trigger."""

    def __init__(self, trigger: TriggerSignal, src: ProcessingElement):
        self._trigger = trigger
        self._src = src
        self._origin: int | None = None          # absolute frame of the latest restart; None = none yet

    def inputs(self) -> list[ProcessingElement]:
        return [self._trigger, self._src]

    def is_pure(self) -> bool:
        return False

    def channel_count(self) -> int | None:
        return self._src.channel_count()

    def resolve_channel_count(self, input_channel_counts: list[int]) -> int:
        if len(input_channel_counts) != 2:
            raise ValueError("TriggerRestartPE expects exactly two inputs")
        return input_channel_counts[1]

    def _compute_extent(self) -> Extent:
        return self._trigger.extent()

    def _forget_origin(self) -> None:
        self._origin = None

    _reset_state = _on_start = _on_stop = _forget_origin

    def _render(self, start: int, duration: int) -> Snippet:
        channels = self.channel_count() or 1
        out = new_output(duration, channels, zero=True)
        fired = np.flatnonzero(self._trigger.render(start, duration).data[:, 0] > 0)
        # the block splits into stretches: [0, first trigger) on the running clock, then one per trigger
        cuts = np.concatenate(([0], fired, [duration])).astype(np.int64)
        for idx in range(len(cuts) - 1):
            row, length = int(cuts[idx]), int(cuts[idx + 1] - cuts[idx])
            restarts = idx > 0
            if length <= 0:
                continue                          # a trigger on frame 0 leaves an empty lead-in
            if restarts:
                self._src.reset_state()
                self._origin = start + row
                piece = self._src.render(0, length)
            elif self._origin is not None:
                piece = self._src.render(start - self._origin, length)
            else:
                continue                          # nothing has started the source yet: silence
            check(lib().pgx_memcpy_d2d(out.offset_ptr(row * channels), piece.dev.ptr, length * channels * 4),
                  "pgx_memcpy_d2d")
        return Snippet(start, out)

    def __repr__(self) -> str:
        return f"TriggerRestartPE(trigger={type(self._trigger).__name__}, src={type(self._src).__name__})"
