"""TriggerSignal: semantic base class for mono integer-valued event streams
(trigger_signal.py:29-121)."""

from __future__ import annotations

import os
from abc import ABC, abstractmethod

import numpy as np

from .processing_element import ProcessingElement
from .snippet import Snippet


def _env_flag(name: str, default: str = "0") -> bool:
    return os.environ.get(name, default).strip().lower() in ("1", "true", "yes", "on")


class TriggerSignal(ProcessingElement, ABC):
    VALIDATE: bool = _env_flag("PYGMU_VALIDATE_SIGNALS", "1")
    VALIDATE_FULL: bool = _env_flag("PYGMU_VALIDATE_SIGNALS_FULL", "0")
    VALIDATE_PROBE_SAMPLES = 64
    ALLOW_MULTIPLE_EVENTS: bool = _env_flag("PYGMU_TRIGGER_ALLOW_MULTIPLE", "1")
    _TRUSTED_DOMAIN = False

    def channel_count(self) -> int:
        return 1

    @abstractmethod
    def _render_trigger(self, start: int, duration: int) -> Snippet:
        raise NotImplementedError

    def _render(self, start: int, duration: int) -> Snippet:
        snip = self._render_trigger(start, duration)
        if self.VALIDATE and not self._TRUSTED_DOMAIN:
            self._validate_trigger_array(snip.data)
        return snip

    @classmethod
    def _validate_trigger_snippet(cls, snip: Snippet) -> None:
        cls._validate_trigger_array(snip.data)

    @classmethod
    def _validate_trigger_array(cls, arr: np.ndarray) -> None:
        if not isinstance(arr, np.ndarray):
            raise TypeError(f"TriggerSignal must render a numpy array, got {type(arr)}")
        if arr.ndim != 2 or arr.shape[1] != 1:
            raise ValueError(f"TriggerSignal must be mono with shape (N,1); got {arr.shape}")
        if arr.dtype.kind not in ("f", "i", "u"):
            raise TypeError(f"TriggerSignal must render numeric dtype; got {arr.dtype}")
        n = arr.shape[0]
        if cls.VALIDATE_FULL or n <= cls.VALIDATE_PROBE_SAMPLES:
            probe = arr[:, 0]
        else:
            idx = np.linspace(0, n - 1, num=min(cls.VALIDATE_PROBE_SAMPLES, n), dtype=int)
            probe = arr[idx, 0]
        if probe.dtype.kind in ("i", "u"):
            vals = probe
        else:
            whole = np.equal(probe, np.round(probe))
            if not np.all(whole):
                bad = probe[~whole]
                raise ValueError("TriggerSignal values must be integers "
                                 f"(found non-integers in probe; min={float(bad.min())}, "
                                 f"max={float(bad.max())}).")
            vals = probe.astype(np.int64)
        if cls.ALLOW_MULTIPLE_EVENTS:
            return
        ok = (vals == -1) | (vals == 0) | (vals == 1)
        if not np.all(ok):
            bad = vals[~ok]
            raise ValueError("TriggerSignal values must be in {-1, 0, +1} "
                             f"(found out-of-domain values in probe; min={int(bad.min())}, "
                             f"max={int(bad.max())}). "
                             "Set PYGMU_TRIGGER_ALLOW_MULTIPLE=1 to allow multiplicity.")
