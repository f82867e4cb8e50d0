"""
EnvelopePE: causal attack/release envelope follower with optional look-ahead
(envelope_pe.py:19-271).

The source is rendered `int(lookahead * sr)` samples ahead, rectified (PEAK) or reduced to
a block-local running RMS (RMS) and smoothed:
  * attack == release: a one-pole low-pass, run as a time-parallel scalar affine scan;
  * otherwise: e += (target > e ? attack_coeff : release_coeff) * (target - e), which
    switches on its own output: solved in windows by time-parallel Newton rounds (k_env_newton).
The per-channel envelope lives in HBM and is zeroed by on_start / on_stop.
"""

from __future__ import annotations

from enum import Enum

import numpy as np

from . import look_ahead as _look_ahead
from ._kernels import DeviceBuffer, check, lib, new_output
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


class DetectionMode(Enum):
    PEAK = "peak"
    RMS = "rms"


class EnvelopePE(ProcessingElement):
    _PASSES_BLOCKS = True              # look_ahead.py: inputs are pulled with the caller's (duration)
    _LOOK_AHEAD_SAFE = True            # look_ahead.py; the RMS detector is block-local (see the condition)
    _STATE_FIELDS = ("_state", "_state_channels")

    def _look_ahead_block_sensitive(self) -> bool:
        # the RMS detector is block-local in the reference (uniform_filter1d restarts at every block edge): a
        # look-ahead window may still render several blocks at once if it tells us where the caller's edges are
        return self._mode == DetectionMode.RMS

    def __init__(self, source: ProcessingElement, attack: float = 0.01, release: float = 0.1,
                 lookahead: float = 0.0, mode: DetectionMode = DetectionMode.PEAK):
        self._source = source
        self._attack = max(0.0, attack)
        self._release = max(0.0, release)
        self._lookahead = max(0.0, min(lookahead, self._attack))
        self._mode = mode
        self._state: DeviceBuffer | None = None
        self._state_channels = 0
        self._scratch_buf: DeviceBuffer | None = None
        self._plan_key = None
        self._scratch_need = 0
        self._coeffs = (1.0, 1.0, 0, 0)

    source = property(lambda self: self._source)
    attack = property(lambda self: self._attack)
    release = property(lambda self: self._release)
    lookahead = property(lambda self: self._lookahead)
    mode = property(lambda self: self._mode)

    def inputs(self) -> list[ProcessingElement]:
        return [self._source]

    def is_pure(self) -> bool:
        return False

    def channel_count(self) -> int | None:
        return self._source.channel_count()

    def _compute_extent(self) -> Extent:
        return self._source.extent()

    def _reset_state(self) -> None:
        if self._state is not None:
            self._state.zero_()

    _on_start = _reset_state
    _on_stop = _reset_state

    def _render(self, start: int, duration: int) -> Snippet:
        sr = self.sample_rate
        src = self._source.render(start + int(self._lookahead * sr), duration)
        ch = src.channels
        if self._state is None or self._state_channels != ch:
            self._state = DeviceBuffer((ch,), np.float64, zero=True)
            self._state_channels = ch
        if self._plan_key != (duration, ch, sr):             # per block shape: scratch size and the coefficients
            self._plan_key = (duration, ch, sr)
            self._scratch_need = lib().pgx_envelope_scratch_bytes(duration, ch)
            # envelope_pe.py:153-158
            a = float(1.0 - np.exp(-1.0 / (self._attack * sr))) if self._attack > 0 else 1.0
            r = float(1.0 - np.exp(-1.0 / (self._release * sr))) if self._release > 0 else 1.0
            window = max(1, int(min(0.01, self._attack) * sr)) if self._mode == DetectionMode.RMS else 0
            self._coeffs = (a, r, 1 if (self._attack == self._release and a < 1.0) else 0, window)
        need = self._scratch_need
        if self._scratch_buf is None or self._scratch_buf.nbytes < need:
            self._scratch_buf = DeviceBuffer((need // 8,), np.float64)
        attack_coeff, release_coeff, one_pole, window = self._coeffs
        out = new_output(duration, ch)
        period = _look_ahead.current_period() if window else 0
        check(lib().pgx_envelope(out.ptr, src.dev.ptr, duration, ch, attack_coeff, release_coeff,
                                 one_pole, window, period, self._state.ptr, self._scratch_buf.ptr),
              "pgx_envelope")
        return Snippet(start, out)

    def __repr__(self) -> str:
        return (f"EnvelopePE(source={type(self._source).__name__}, attack={self._attack}, "
                f"release={self._release}, lookahead={self._lookahead}, mode={self._mode.value})")
