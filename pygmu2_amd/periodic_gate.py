"""
PeriodicGate: periodic rectangular 0/1 gate (periodic_gate.py:30-67 over the pure
rectangle path of FunctionGenPE, function_gen_pe.py:157-193).

Scalar frequency / duty / phase -> pure, generated on the device with the reference's
float64 expression  phase = mod(mod(n * (f/sr), 1) + phase, 1);  gate = phase < duty.
Any PE-driven parameter -> FunctionGenPE's stateful path (function_gen_pe.py:169-176): the phase is the
running sum of f/sr, carried across contiguous renders and restarted from 0 otherwise; the sum is a
prefix scan on the device (pgx_gate_stateful), so a gate sample may differ from the reference's
sequential np.cumsum only where the phase meets the duty threshold to ~1e-13.
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
    _LOOK_AHEAD_SAFE = True            # PE-driven parameters: carried phase
    _STATE_FIELDS = ("_state", "_last_render_end")

    _TRUSTED_DOMAIN = True          # the kernel emits exactly 0.0f or 1.0f

    def __init__(self, frequency=1.0, duty_cycle=0.5, phase=0.0):
        self._frequency = frequency
        self._duty_cycle = duty_cycle
        self._phase = phase
        self._params: DeviceBuffer | None = None
        self._state: DeviceBuffer | None = None       # stateful path: carried phase
        self._last_render_end: int | None = None

    def _pe_params(self) -> list[ProcessingElement]:
        return [p for p in (self._frequency, self._duty_cycle, self._phase) if isinstance(p, ProcessingElement)]

    def inputs(self) -> list[ProcessingElement]:
        return self._pe_params()

    def is_pure(self) -> bool:
        return not self._pe_params()

    def _compute_extent(self) -> Extent:
        ext = Extent(None, None)
        for p in self._pe_params():
            ext = ext.intersection(p.extent())
        return ext

    def _reset_state(self) -> None:
        if self._state is not None:
            self._state.zero_()
        self._last_render_end = None

    _on_start = _reset_state
    _on_stop = _reset_state

    def _gate_params(self) -> dict:
        if self._pe_params():
            raise TypeError("PeriodicGate with PE parameters has no closed form")
        dt = np.float64(self._frequency) / float(self.sample_rate)       # freq / sr (:162)
        duty = float(np.clip(self._duty_cycle, 0.0, 1.0))                 # (:177)
        return dict(dt=float(dt), phase=self._phase, duty=duty)

    def _render_gate(self, start: int, duration: int) -> Snippet:
        if self._pe_params():
            return self._render_stateful(start, duration)
        if self._params is None:
            self._params = _dev.upload_struct(_dev.GATE_PARAMS, **self._gate_params())
        out = new_output(duration, 1)
        check(lib().pgx_periodic_gate(out.ptr, 0, 1, start, duration, self._params.ptr),
              "pgx_periodic_gate")
        return Snippet(start, out)

    def _render_stateful(self, start: int, duration: int) -> Snippet:
        if self._state is None:
            self._state = DeviceBuffer((1,), np.float64, zero=True)
        if self._last_render_end is None or start != self._last_render_end:
            self._state.zero_()
        f_s, f_buf = self._control_stream(self._frequency, start, duration)
        d_s, d_buf = self._control_stream(self._duty_cycle, start, duration)
        p_s, p_buf = self._control_stream(self._phase, start, duration)
        out = new_output(duration, 1)
        check(lib().pgx_gate_stateful(out.ptr, duration, float(self.sample_rate),
                                      0.0 if f_s is None else f_s, 0.0 if d_s is None else d_s,
                                      0.0 if p_s is None else p_s, None if f_buf is None else f_buf.ptr,
                                      None if d_buf is None else d_buf.ptr, None if p_buf is None else p_buf.ptr,
                                      self._state.ptr), "pgx_gate_stateful")
        self._last_render_end = start + duration
        return Snippet(start, out)
