"""
SinePE: sine oscillator with scalar or PE-driven frequency / amplitude / phase
(sine_pe.py:51-270).

All-scalar parameters -> pure: phase is computed directly from the sample index in
float64 on the device (pgx_sine_render).  Any PE parameter -> stateful: the frequency is
integrated by a device prefix scan and the accumulated phase is carried between
contiguous renders in a device-resident state blob (pgx_sine_stateful).
"""

from __future__ import annotations

import numpy as np

from . import device as _dev
from ._kernels import DeviceBuffer, check, lib, new_output, ptr
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


class SinePE(ProcessingElement):
    _READ_AHEAD_SAFE = True
    _LOOK_AHEAD_SAFE = True            # stateful path: the carried phase is sample-exact ...
    _STATE_FIELDS = ("_state",)

    def _look_ahead_condition(self) -> bool:
        # ... unless a phase offset is in play: the reference carries `phase + offset` into the next block
        # and adds the offset again (sine_pe.py:217-232), so its output depends on where the blocks are cut
        return self.is_pure() or (not isinstance(self._phase, ProcessingElement) and float(self._phase) == 0.0)

    def __init__(self, frequency=440.0, amplitude=1.0, phase=0.0, channels: int = 1):
        self._frequency = frequency
        self._amplitude = amplitude
        self._phase = phase
        self._channels = channels
        self._params: DeviceBuffer | None = None      # uploaded once, lazily
        self._state: DeviceBuffer | None = None       # {accumulated_phase, initialised}

    frequency = property(lambda self: self._frequency)
    amplitude = property(lambda self: self._amplitude)
    initial_phase = property(lambda self: self._phase)

    def _has_pe_inputs(self) -> bool:
        return any(isinstance(p, ProcessingElement)
                   for p in (self._frequency, self._amplitude, self._phase))

    def inputs(self) -> list[ProcessingElement]:
        return [p for p in (self._frequency, self._amplitude, self._phase)
                if isinstance(p, ProcessingElement)]

    def is_pure(self) -> bool:
        return not self._has_pe_inputs()

    def channel_count(self) -> int:
        return self._channels

    def _compute_extent(self) -> Extent:
        ext = Extent(None, None)
        for pe in self.inputs():
            ext = ext.intersection(pe.extent())
        return ext

    def _clear_phase(self) -> None:
        if self._state is not None:
            self._state.zero_()

    # the reference resets the accumulated phase in _on_start/_on_stop only (sine_pe.py:109-116): there is
    # no _reset_state hook, so reset_state() -- e.g. from TriggerRestartPE -- leaves a stateful SinePE running
    _on_start = _clear_phase
    _on_stop = _clear_phase

    def _pure_params(self) -> DeviceBuffer:
        if self._params is None:
            # same host arithmetic as the reference: (2.0 * pi) * f, left to right
            w = 2.0 * np.pi * float(self._frequency)
            self._params = _dev.upload_struct(_dev.SINE_PARAMS, w=w, amp=float(self._amplitude),
                                              phase0=float(self._phase))
        return self._params

    def _render_with_gain(self, start: int, duration: int, gain32: float) -> Snippet:
        """GainPE(self, gain=<scalar>) in one launch (pure path only); same roundings as two PEs."""
        out = new_output(duration, self._channels)
        check(lib().pgx_sine_gain_render(out.ptr, 0, 1, start, duration, self._channels,
                                         float(self.sample_rate), self._pure_params().ptr, gain32),
              "pgx_sine_gain_render")
        return Snippet(start, out)

    def _render(self, start: int, duration: int) -> Snippet:
        out = new_output(duration, self._channels)
        L = lib()
        if not self._has_pe_inputs():
            check(L.pgx_sine_render(out.ptr, 0, 1, start, duration, self._channels,
                                    float(self.sample_rate), self._pure_params().ptr), "pgx_sine_render")
            return Snippet(start, out)

        f_s, f_buf = self._control_stream(self._frequency, start, duration)
        a_s, a_buf = self._control_stream(self._amplitude, start, duration)
        p_s, p_buf = self._control_stream(self._phase, start, duration)
        if self._params is None:
            self._params = _dev.upload_struct(
                _dev.SINE_STATEFUL_PARAMS, freq=0.0 if f_s is None else f_s,
                amp=0.0 if a_s is None else a_s, phase=0.0 if p_s is None else p_s,
                phase_is_stream=int(p_buf is not None))
        if self._state is None:
            self._state = DeviceBuffer((2,), np.float64, zero=True)
        check(L.pgx_sine_stateful(out.ptr, duration, self._channels, float(self.sample_rate),
                                  self._params.ptr, ptr(f_buf), ptr(a_buf), ptr(p_buf),
                                  self._state.ptr), "pgx_sine_stateful")
        return Snippet(start, out)

    def __repr__(self) -> str:
        def s(p):
            return type(p).__name__ if isinstance(p, ProcessingElement) else str(p)
        return (f"SinePE(frequency={s(self._frequency)}, amplitude={s(self._amplitude)}, "
                f"phase={s(self._phase)}, channels={self._channels})")
