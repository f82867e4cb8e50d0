"""PeriodicTrigger: +amplitude impulses every round(sr/hz) samples
(periodic_trigger.py:25-58).  Integer arithmetic -> bit-exact."""

from __future__ import annotations

from ._kernels import check, lib, new_output
from .config import get_sample_rate
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet
from .trigger_signal import TriggerSignal


class PeriodicTrigger(TriggerSignal):
    _READ_AHEAD_SAFE = True

    _TRUSTED_DOMAIN = True

    def __init__(self, hz: float, phase: float = 0.0, amplitude: int = 1):
        if hz <= 0:
            raise ValueError("PeriodicTrigger hz must be > 0")
        rate, cycle = float(hz), float(phase) % 1.0
        period = int(round(get_sample_rate() / rate))                 # samples between two events
        if period <= 0:
            raise ValueError("PeriodicTrigger computed period <= 0; check sample rate / hz")
        self._hz, self._phase, self._amp, self._period = rate, cycle, int(amplitude), period
        self._phase_samples = int(round(cycle * period))              # the kernel fires where (n + this) % period == 0

    def inputs(self) -> list[ProcessingElement]:
        return []

    def is_pure(self) -> bool:
        return True

    def _compute_extent(self) -> Extent:
        return Extent(None, None)

    def _render_trigger(self, start: int, duration: int) -> Snippet:
        out = new_output(duration, 1)
        check(lib().pgx_periodic_trigger(out.ptr, start, duration, self._period, self._phase_samples,
                                         float(self._amp)), "pgx_periodic_trigger")
        return Snippet(start, out)
