"""
PeriodicGate: periodic rectangular 0/1 gate (periodic_gate.py:30-67 over the pure
rectangle path of FunctionGenPE, function_gen_pe.py:157-193).

Scalar frequency / duty / phase -> pure, generated on the device with the reference's
float64 expression  phase = mod(mod(n * (f/sr), 1) + phase, 1);  gate = phase < duty.
PE-driven parameters (the reference's stateful FunctionGenPE path) are outside the
accelerated hot path (SURVEY.md section 8 a16 covers the scalar case) and raise.
"""

from __future__ import annotations

import numpy as np

from . import device as _dev
from ._kernels import DeviceBuffer, check, lib, new_output
from .extent import Extent
from .gate_signal import GateSignal
from .processing_element import ProcessingElement
from .snippet import Snippet


class PeriodicGate(GateSignal):
    _READ_AHEAD_SAFE = True

    _TRUSTED_DOMAIN = True          # the kernel emits exactly 0.0f or 1.0f

    def __init__(self, frequency=1.0, duty_cycle=0.5, phase=0.0):
        for name, p in (("frequency", frequency), ("duty_cycle", duty_cycle), ("phase", phase)):
            if isinstance(p, ProcessingElement):
                raise NotImplementedError(
                    f"PeriodicGate({name}=<PE>) is not on the accelerated render path; "
                    "only scalar parameters are supported")
        self._frequency = float(frequency)
        self._duty_cycle = float(duty_cycle)
        self._phase = float(phase)
        self._params: DeviceBuffer | None = None

    def inputs(self) -> list[ProcessingElement]:
        return []

    def is_pure(self) -> bool:
        return True

    def _compute_extent(self) -> Extent:
        return Extent(None, None)

    def _gate_params(self) -> dict:
        dt = np.float64(self._frequency) / float(self.sample_rate)       # freq / sr (:162)
        duty = float(np.clip(self._duty_cycle, 0.0, 1.0))                 # (:177)
        return dict(dt=float(dt), phase=self._phase, duty=duty)

    def _render_gate(self, start: int, duration: int) -> Snippet:
        if self._params is None:
            self._params = _dev.upload_struct(_dev.GATE_PARAMS, **self._gate_params())
        out = new_output(duration, 1)
        check(lib().pgx_periodic_gate(out.ptr, 0, 1, start, duration, self._params.ptr),
              "pgx_periodic_gate")
        return Snippet(start, out)
