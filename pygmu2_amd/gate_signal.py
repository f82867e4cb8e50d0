"""GateSignal: semantic base class for mono 0/1 control signals (gate_signal.py:28-99).
Subclasses implement `_render_gate`; the optional validation probes up to 64 evenly spaced
samples (or all of them with PYGMU_VALIDATE_SIGNALS_FULL=1) on the host."""

from __future__ import annotations

import os
from abc import ABC, abstractmethod

import numpy as np

from .processing_element import ProcessingElement
from .snippet import Snippet


def _env_flag(name: str, default: str = "0") -> bool:
    return os.environ.get(name, default).strip().lower() in ("1", "true", "yes", "on")


class GateSignal(ProcessingElement, ABC):
    VALIDATE: bool = _env_flag("PYGMU_VALIDATE_SIGNALS", "1")
    VALIDATE_FULL: bool = _env_flag("PYGMU_VALIDATE_SIGNALS_FULL", "0")
    VALIDATE_PROBE_SAMPLES = 64
    # Device-generated gates are 0/1 by construction; probing them would force a
    # device->host copy per render, so subclasses whose kernel guarantees the domain set this.
    _TRUSTED_DOMAIN = False

    def channel_count(self) -> int:
        return 1

    @abstractmethod
    def _render_gate(self, start: int, duration: int) -> Snippet:
        raise NotImplementedError

    def _render(self, start: int, duration: int) -> Snippet:
        snip = self._render_gate(start, duration)
        if self.VALIDATE and not self._TRUSTED_DOMAIN:
            self._validate_gate_array(snip.data)
        return snip

    @classmethod
    def _validate_gate_snippet(cls, snip: Snippet) -> None:
        cls._validate_gate_array(snip.data)

    @classmethod
    def _validate_gate_array(cls, arr: np.ndarray) -> None:
        if not isinstance(arr, np.ndarray):
            raise TypeError(f"GateSignal must render a numpy array, got {type(arr)}")
        if arr.ndim != 2 or arr.shape[1] != 1:
            raise ValueError(f"GateSignal must be mono with shape (N,1); got {arr.shape}")
        if arr.dtype.kind not in ("f", "i", "u"):
            raise TypeError(f"GateSignal must render numeric dtype; got {arr.dtype}")
        n = arr.shape[0]
        if cls.VALIDATE_FULL or n <= cls.VALIDATE_PROBE_SAMPLES:
            probe = arr[:, 0]
        else:
            idx = np.linspace(0, n - 1, num=min(cls.VALIDATE_PROBE_SAMPLES, n), dtype=int)
            probe = arr[idx, 0]
        ok = (probe == 0.0) | (probe == 1.0)
        if not np.all(ok):
            bad = probe[~ok]
            raise ValueError(
                "GateSignal values must be exactly 0 or 1 "
                f"(found out-of-domain values in probe; min={float(bad.min())}, max={float(bad.max())}). "
                "If you meant to threshold a control/audio signal, wrap it with ToGateSignal.")
