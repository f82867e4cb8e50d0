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
    def __init__(self, trigger: TriggerSignal, src: ProcessingElement):
        self._trigger = trigger
        self._src = src
        self._t0_abs: int | None = None

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

    def _reset_state(self) -> None:
        self._t0_abs = None

    _on_start = _reset_state
    _on_stop = _reset_state

    def _render(self, start: int, duration: int) -> Snippet:
        n = duration
        ch = self.channel_count() or 1
        out = new_output(n, ch, zero=True)
        trig = self._trigger.render(start, duration).data[:, 0]
        events = np.nonzero(trig > 0)[0]

        def place(row: int, snip: Snippet) -> None:
            if snip.duration:
                check(lib().pgx_memcpy_d2d(out.offset_ptr(row * ch), snip.dev.ptr, snip.duration * ch * 4),
                      "pgx_memcpy_d2d")

        prefix_end = int(events[0]) if events.size else n
        if prefix_end > 0 and self._t0_abs is not None:
            place(0, self._src.render(start - self._t0_abs, prefix_end))
        for i, k in enumerate(events.tolist()):
            k_end = int(events[i + 1]) if i + 1 < events.size else n
            if k_end <= k:
                continue
            self._src.reset_state()
            self._t0_abs = start + k
            place(k, self._src.render(0, k_end - k))
        return Snippet(start, out)

    def __repr__(self) -> str:
        return f"TriggerRestartPE(trigger={type(self._trigger).__name__}, src={type(self._src).__name__})"
