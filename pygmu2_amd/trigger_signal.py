"""TriggerSignal: semantic base class for mono integer-valued event streams
(trigger_signal.py:29-121).  Validation: see _signal_domain.py."""

from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np

from . import _signal_domain as _domain
from .processing_element import ProcessingElement
from .snippet import Snippet


class TriggerSignal(ProcessingElement, ABC):
    VALIDATE: bool = _domain.env_flag("PYGMU_VALIDATE_SIGNALS", "1")
    VALIDATE_FULL: bool = _domain.env_flag("PYGMU_VALIDATE_SIGNALS_FULL", "0")
    VALIDATE_PROBE_SAMPLES = _domain.PROBE_SAMPLES
    ALLOW_MULTIPLE_EVENTS: bool = _domain.env_flag("PYGMU_TRIGGER_ALLOW_MULTIPLE", "1")
    _TRUSTED_DOMAIN = False

    def channel_count(self) -> int:
        return 1

    @abstractmethod
    def _render_trigger(self, start: int, duration: int) -> Snippet:
        raise NotImplementedError

    def _render(self, start: int, duration: int) -> Snippet:
        events = self._render_trigger(start, duration)
        if self.VALIDATE and not self._TRUSTED_DOMAIN:
            self._validate_trigger_array(events.data)
        return events

    @classmethod
    def _validate_trigger_snippet(cls, snip: Snippet) -> None:
        cls._validate_trigger_array(snip.data)

    @classmethod
    def _validate_trigger_array(cls, arr: np.ndarray) -> None:
        seen = _domain.probe_column("TriggerSignal", arr, cls.VALIDATE_FULL)
        if seen.dtype.kind == "f":
            fractional = seen[seen != np.round(seen)]
            if fractional.size:
                lo, hi = _domain.span(fractional)
                raise ValueError("TriggerSignal values must be integers "
                                 f"(found non-integers in probe; min={float(lo)}, max={float(hi)}).")
            seen = seen.astype(np.int64)
        if cls.ALLOW_MULTIPLE_EVENTS:
            return
        outside = seen[np.abs(seen) > 1]
        if outside.size:
            lo, hi = _domain.span(outside)
            raise ValueError("TriggerSignal values must be in {-1, 0, +1} "
                             f"(found out-of-domain values in probe; min={int(lo)}, max={int(hi)}). "
                             "Set PYGMU_TRIGGER_ALLOW_MULTIPLE=1 to allow multiplicity.")
