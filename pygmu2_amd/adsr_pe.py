"""
AdsrGatedPE / AdsrTriggeredPE: linear-segment ADSR envelopes (adsr_pe.py:58-335).

Per sample: emit the current level, then react to the gate edge / trigger, then advance
the active segment by its per-sample slope with clamping.  The float64 accumulation is
reproduced step by step on the device (one lane per envelope), so the float32 output is
bit-identical to the reference's Python loop.  State {segment, level, previous gate |
sustain deadline} lives in HBM.
"""

from __future__ import annotations

import numpy as np

from . import device as _dev
from ._kernels import DeviceBuffer, check, lib, new_output
from .config import get_sample_rate
from .extent import Extent
from .gate_signal import GateSignal
from .processing_element import ProcessingElement
from .snippet import Snippet
from .trigger_signal import TriggerSignal

IDLE, ATTACK, DECAY, SUSTAIN, RELEASE = "idle", "attack", "decay", "sustain", "release"


def _slopes(attack_time, decay_time, sustain_level, release_time):
    """Per-sample increments, evaluated like the reference (adsr_pe.py:78-81)."""
    sr = float(get_sample_rate())
    return ((1.0 - 0.0) / (attack_time * sr),
            (sustain_level - 1.0) / (decay_time * sr),
            (0.0 - sustain_level) / (release_time * sr))


class _AdsrBase(ProcessingElement):
    _LOOK_AHEAD_SAFE = True            # look_ahead.py
    _STATE_FIELDS = ("_state",)

    def _init_common(self, control, attack_time, decay_time, sustain_level, release_time):
        self._control = control
        self._attack_time = float(attack_time)
        self._decay_time = float(decay_time)
        self._sustain_level = float(sustain_level)
        self._release_time = float(release_time)
        self._attack_dvdt, self._decay_dvdt, self._release_dvdt = _slopes(
            self._attack_time, self._decay_time, self._sustain_level, self._release_time)
        self._params: DeviceBuffer | None = None
        self._state: DeviceBuffer | None = None
        self._workspace: DeviceBuffer | None = None

    def is_pure(self) -> bool:
        return False

    def channel_count(self) -> int:
        return 1

    def inputs(self) -> list[ProcessingElement]:
        return [self._control]

    def _compute_extent(self) -> Extent:
        return self._control.extent()

    def _reset_state(self) -> None:
        if self._state is not None:
            self._state.zero_()

    _on_start = _reset_state
    _on_stop = _reset_state

    def _adsr_params(self) -> dict:
        return dict(attack_dvdt=self._attack_dvdt, decay_dvdt=self._decay_dvdt,
                    release_dvdt=self._release_dvdt, sustain_level=self._sustain_level,
                    sustain_samples=getattr(self, "_sustain_samples", 0))

    def _ensure_device(self) -> None:
        if self._params is None:
            self._params = _dev.upload_struct(_dev.ADSR_PARAMS, **self._adsr_params())
        if self._state is None:
            self._state = DeviceBuffer((3,), np.float64, zero=True)

    def _scratch(self, duration: int) -> DeviceBuffer:
        need = lib().pgx_adsr_workspace_bytes(1, duration)
        if self._workspace is None or self._workspace.nbytes < need:
            self._workspace = DeviceBuffer((need,), np.uint8)
        return self._workspace

    def _control_mono(self, start: int, duration: int) -> DeviceBuffer:
        snip = self._control.render(start=start, duration=duration)
        if snip.channels != 1:
            raise ValueError(f"ADSR control input must be mono, got {snip.channels} channels")
        return snip.dev


class AdsrGatedPE(_AdsrBase):
    def __init__(self, gate: GateSignal, attack_time: float = 0.1, decay_time: float = 0.1,
                 sustain_level: float = 0.5, release_time: float = 0.1):
        self._init_common(gate, attack_time, decay_time, sustain_level, release_time)
        self._gate = gate

    def _render(self, start: int, duration: int) -> Snippet:
        self._ensure_device()
        gate = self._control_mono(start, duration)
        out = new_output(duration, 1)
        check(lib().pgx_adsr_gated(out.ptr, 0, gate.ptr, 0, 1, duration, self._params.ptr,
                                   self._state.ptr, self._scratch(duration).ptr), "pgx_adsr_gated")
        return Snippet(start, out)


class AdsrTriggeredPE(_AdsrBase):
    def __init__(self, trigger: TriggerSignal, attack_time: float = 0.1, decay_time: float = 0.1,
                 sustain_time: float = 0.5, sustain_level: float = 0.5, release_time: float = 0.1):
        self._sustain_time = float(sustain_time)
        self._sustain_samples = int(round(self._sustain_time * float(get_sample_rate())))
        self._init_common(trigger, attack_time, decay_time, sustain_level, release_time)
        self._trigger = trigger

    def _render(self, start: int, duration: int) -> Snippet:
        self._ensure_device()
        trig = self._control_mono(start, duration)
        out = new_output(duration, 1)
        check(lib().pgx_adsr_triggered(out.ptr, 0, trig.ptr, 0, 1, start, duration, self._params.ptr,
                                       self._state.ptr, self._scratch(duration).ptr), "pgx_adsr_triggered")
        return Snippet(start, out)
