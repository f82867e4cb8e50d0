"""
SuperSawPE: detuned unison of BlitSaw voices (super_saw_pe.py:77-342).

Voice layout, detune ratios, mix gains (equal / linear / center_heavy, normalised to
unit power) and the seeded random initial phases follow the reference.  Rendering is two
launches regardless of the voice count: one batched pgx_blitsaw over all voices (one
workgroup per voice) and one pgx_supersaw_sum that accumulates the float32 voices in
float64 in voice order, applies the amplitude and rounds to float32 -- the same
roundings as the reference's per-voice Python loop.
"""

from __future__ import annotations

import numpy as np

from . import device as _dev
from ._kernels import DeviceBuffer, blitsaw_workspace, check, lib, new_output, ptr
from . import blit_saw_pe as _blit
from .blit_saw_pe import BlitSawPE
from .cache_pe import CachePE
from .extent import Extent
from .gain_pe import GainPE
from .processing_element import ProcessingElement
from .snippet import Snippet


class SuperSawPE(ProcessingElement):
    _LOOK_AHEAD_SAFE = True            # look_ahead.py
    _STATE_FIELDS = ("_state", "_last_render_end")

    MIX_EQUAL = "equal"
    MIX_CENTER_HEAVY = "center_heavy"
    MIX_LINEAR = "linear"

    def __init__(self, frequency, amplitude=1.0, voices: int = 7, detune_cents: float = 20.0,
                 mix_mode: str = "center_heavy", channels: int = 1, randomize_phase: bool = True,
                 seed: int | None = None):
        if voices < 1:
            voices = 1
        self._frequency = frequency
        self._amplitude = amplitude
        self._voices = voices
        self._detune_cents = detune_cents
        self._mix_mode = mix_mode
        self._channels = channels
        self._randomize_phase = bool(randomize_phase)
        self._rng = np.random.default_rng(seed)
        self._detune_ratios = self._compute_detune_ratios()
        self._mix_gains = self._compute_mix_gains()
        self._oscillators: list[BlitSawPE] = self._create_oscillators()
        # device-side voice bank
        self._params: DeviceBuffer | None = None
        self._state: DeviceBuffer | None = None      # [V][2]
        self._amp_scalar: DeviceBuffer | None = None
        self._last_render_end: int | None = None

    frequency = property(lambda self: self._frequency)
    amplitude = property(lambda self: self._amplitude)
    voices = property(lambda self: self._voices)
    detune_cents = property(lambda self: self._detune_cents)
    mix_mode = property(lambda self: self._mix_mode)

    # ------------------------------------------------------------------ voice tables
    def _compute_detune_ratios(self) -> np.ndarray:
        if self._voices == 1 or self._detune_cents == 0:
            return np.array([1.0])
        cents = np.linspace(-self._detune_cents, self._detune_cents, self._voices)
        return 2 ** (cents / 1200.0)

    def _compute_mix_gains(self) -> np.ndarray:
        n, mode = self._voices, self._mix_mode
        if n <= 0:
            raise ValueError("n must be >= 1")
        if n == 1:
            return np.array([1.0])
        gains = np.ones(n, dtype=np.float32)
        if mode == self.MIX_EQUAL:
            pass
        elif mode == self.MIX_LINEAR:
            center = (n - 1) / 2.0
            dist = np.abs(np.arange(n, dtype=np.float32) - center)
            gains = 0.5 + 0.5 * (1.0 - dist / np.max(dist))
        elif mode == self.MIX_CENTER_HEAVY:
            gains[:] = 0.5
            if n % 2 == 1:
                gains[n // 2] = 1.0
            else:
                gains[n // 2 - 1] = 1.0
                gains[n // 2] = 1.0
        else:
            raise ValueError(f"Unknown mix mode: {mode}")
        return gains / np.sqrt(np.sum(gains ** 2))

    def _create_oscillators(self) -> list[BlitSawPE]:
        oscs = []
        if isinstance(self._frequency, ProcessingElement):
            CachePE(self._frequency)          # the reference builds (and never uses) this wrapper
        for i, ratio in enumerate(self._detune_ratios):
            if isinstance(self._frequency, ProcessingElement):
                freq = GainPE(self._frequency, gain=ratio)
            else:
                freq = self._frequency * ratio
            oscs.append(BlitSawPE(frequency=freq, amplitude=self._mix_gains[i], channels=1,
                                  initial_phase=self._rng.random(1) if self._randomize_phase else 0.0))
        return oscs

    # ------------------------------------------------------------------ PE contract
    def inputs(self) -> list[ProcessingElement]:
        return [p for p in (self._frequency, self._amplitude) if isinstance(p, ProcessingElement)]

    def is_pure(self) -> bool:
        return False

    def channel_count(self) -> int:
        return self._channels

    def _compute_extent(self) -> Extent:
        ext = Extent(None, None)
        for pe in self.inputs():
            ext = ext.intersection(pe.extent())
        return ext

    def _reset_state(self) -> None:
        self._last_render_end = None
        for osc in self._oscillators:
            osc.reset_state()

    def _on_start(self) -> None:
        self._reset_state()

    def _on_stop(self) -> None:
        self._last_render_end = None
        for osc in self._oscillators:
            osc.on_stop()

    # ------------------------------------------------------------------ device tables
    def _voice_param_records(self) -> np.ndarray:
        rec = np.zeros(len(self._oscillators), dtype=_dev.BLITSAW_PARAMS)
        for i, osc in enumerate(self._oscillators):
            p = osc._scalar_params()
            for k, v in p.items():
                rec[i][k] = v
        return rec

    _wide_records = _voice_param_records

    def _voice_initial_state(self) -> np.ndarray:
        return np.stack([osc._initial_state() for osc in self._oscillators])

    def _render(self, start: int, duration: int) -> Snippet:
        L = lib()
        nv = len(self._oscillators)
        sr = float(self.sample_rate)
        if self._params is None:
            self._params = _dev.upload_structs(self._voice_param_records())
        if self._state is None:
            self._state = DeviceBuffer((nv, 2), np.float64)
            self._last_render_end = None
        if self._last_render_end is None or start != self._last_render_end:
            self._state.upload(self._voice_initial_state())

        if (_blit.WIDE_LONG_RENDERS and duration >= _blit.WIDE_MIN_FRAMES and not self.inputs() and nv <= 16):
            # a long block (a look-ahead window) of a scalar-parameter SuperSaw: voices summed on chip, time segments
            out = _blit.render_wide(self, nv, start, duration, float(self._amplitude), self._channels)
            if out is not None:
                self._last_render_end = start + duration
                return Snippet(start, out)

        # per-voice frequency streams for a PE frequency: GainPE(freq, ratio) == f32 * f32(ratio)
        f_buf, f_stride = None, 0
        if isinstance(self._frequency, ProcessingElement):
            _, base = self._control_stream(self._frequency, start, duration)
            f_buf = DeviceBuffer((nv, duration), np.float32)
            for i, ratio in enumerate(self._detune_ratios):
                check(L.pgx_gain_const(f_buf.offset_ptr(i * duration), base.ptr, duration,
                                       float(np.float32(ratio))), "pgx_gain_const")
            f_stride = duration
        voices = DeviceBuffer((nv, duration), np.float32)
        ws = blitsaw_workspace(self, nv, duration, f_buf is not None)
        check(L.pgx_blitsaw(voices.ptr, duration, nv, duration, 1, sr, self._params.ptr,
                            ptr(f_buf), f_stride, None, 0, None, 0, self._state.ptr, ptr(ws), None), "pgx_blitsaw")

        a_s, a_buf = self._control_stream(self._amplitude, start, duration)
        if self._amp_scalar is None:
            self._amp_scalar = DeviceBuffer.from_host(np.array([0.0 if a_s is None else a_s]))
        out = new_output(duration, self._channels)
        check(L.pgx_supersaw_sum(out.ptr, 0, 1, nv, duration, self._channels, voices.ptr,
                                 self._amp_scalar.ptr, ptr(a_buf), 0), "pgx_supersaw_sum")
        self._last_render_end = start + duration
        return Snippet(start, out)

    def __repr__(self) -> str:
        f = (type(self._frequency).__name__ if isinstance(self._frequency, ProcessingElement)
             else str(self._frequency))
        return (f"SuperSawPE(frequency={f}, voices={self._voices}, "
                f"detune_cents={self._detune_cents}, mix_mode={self._mix_mode!r})")
