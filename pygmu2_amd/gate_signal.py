"""GateSignal: semantic base class for mono 0/1 control signals (gate_signal.py:28-99).
Subclasses implement `_render_gate`; the optional validation (see _signal_domain.py) runs on the host."""

from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np

from . import _signal_domain as _domain
from .processing_element import ProcessingElement
from .snippet import Snippet


class GateSignal(ProcessingElement, ABC):
    VALIDATE: bool = _domain.env_flag("PYGMU_VALIDATE_SIGNALS", "1")
    VALIDATE_FULL: bool = _domain.env_flag("PYGMU_VALIDATE_SIGNALS_FULL", "0")
    VALIDATE_PROBE_SAMPLES = _domain.PROBE_SAMPLES
    # Device-generated gates are 0/1 by construction; probing them would force a
    # device->host copy per render, so subclasses whose kernel guarantees the domain set this.
    _TRUSTED_DOMAIN = False

    def channel_count(self) -> int:
        return 1

    @abstractmethod
    def _render_gate(self, start: int, duration: int) -> Snippet:
        raise NotImplementedError

    def _render(self, start: int, duration: int) -> Snippet:
        gate = self._render_gate(start, duration)
        if self.VALIDATE and not self._TRUSTED_DOMAIN:
            self._validate_gate_array(gate.data)
        return gate

    @classmethod
    def _validate_gate_snippet(cls, snip: Snippet) -> None:
        cls._validate_gate_array(snip.data)

    @classmethod
    def _validate_gate_array(cls, arr: np.ndarray) -> None:
        seen = _domain.probe_column("GateSignal", arr, cls.VALIDATE_FULL)
        outside = seen[(seen != 0.0) & (seen != 1.0)]
        if outside.size:
            lo, hi = _domain.span(outside)
            raise ValueError(
                "GateSignal values must be exactly 0 or 1 "
                f"(found out-of-domain values in probe; min={float(lo)}, max={float(hi)}). "
                "If you meant to threshold a control/audio signal, wrap it with ToGateSignal.")
