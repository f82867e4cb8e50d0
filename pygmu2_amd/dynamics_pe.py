"""
DynamicsPE: compressor / limiter / expander / gate driven by an external envelope (dynamics_pe.py:25-386).

  level_db = 20 log10(max(envelope, 1e-10));  gain_db = curve(level_db) + makeup;  out = audio * 10**(gain_db/20)

numpy evaluates all of it in float32 (the envelope is float32, Python scalars are weak), so the scalars are
rounded to float32 here exactly where numpy's promotion does and the device follows with float32 arithmetic
(pgx_dynamics).  The automatic make-up gain is host arithmetic (float64), as in the reference.
"""

from __future__ import annotations

import math
from enum import Enum

import numpy as np

from . import device as _dev
from ._kernels import DeviceBuffer, check, lib, new_output
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


class DynamicsMode(Enum):
    COMPRESS = "compress"
    EXPAND = "expand"
    LIMIT = "limit"
    GATE = "gate"


def db_to_ratio(db):
    """conversions.py:141-160: 10 ** (dB / 20)."""
    return np.power(10.0, np.asarray(db, dtype=np.float64) / 20.0)


def ratio_to_db(ratio):
    """conversions.py:112-138: 20 log10(ratio)."""
    return 20.0 * np.log10(np.asarray(ratio, dtype=np.float64))


class DynamicsPE(ProcessingElement):
    _PASSES_BLOCKS = True              # look_ahead.py: inputs are pulled with the caller's (duration)
    _LOOK_AHEAD_SAFE = True            # look_ahead.py: an element-wise function of its two inputs, no state
    AUTO = "auto"

    def __init__(self, source: ProcessingElement, envelope: ProcessingElement, threshold: float = -20.0,
                 ratio: float = 4.0, knee: float = 0.0, makeup_gain: float | str = "auto",
                 mode: DynamicsMode = DynamicsMode.COMPRESS, stereo_link: bool = True, gate_range: float = -80.0):
        self._source = source
        self._envelope = envelope
        self._threshold = threshold
        self._ratio = max(0.001, ratio)
        self._knee = max(0.0, knee)
        self._makeup_gain = makeup_gain
        self._mode = mode
        self._stereo_link = stereo_link
        self._range = gate_range
        # The reference's automatic value for a compressor / limiter is a numpy float64 scalar (a strong type):
        # adding it promotes the float32 gain curve to float64.  An explicit value is a Python float (weak).
        self._wide_makeup = makeup_gain == self.AUTO and mode in (DynamicsMode.COMPRESS, DynamicsMode.LIMIT)
        if makeup_gain == self.AUTO:
            self._makeup_gain_db = self._compute_auto_makeup()
        else:
            self._makeup_gain_db = float(makeup_gain)
        self._makeup_gain_linear = db_to_ratio(self._makeup_gain_db)
        self._params: DeviceBuffer | None = None

    threshold = property(lambda self: self._threshold)
    ratio = property(lambda self: self._ratio)
    knee = property(lambda self: self._knee)
    makeup_gain = property(lambda self: self._makeup_gain_db)
    mode = property(lambda self: self._mode)
    stereo_link = property(lambda self: self._stereo_link)

    def inputs(self) -> list[ProcessingElement]:
        return [self._source, self._envelope]

    def is_pure(self) -> bool:
        return True

    def channel_count(self) -> int | None:
        return self._source.channel_count()

    def _compute_extent(self) -> Extent:
        return self._source.extent().intersection(self._envelope.extent())

    # ------------------------------------------------------------------ scalar gain curve (float64, host)
    def _compute_gain_db(self, level_db: float) -> float:
        """dynamics_pe.py:190-325 for one level (used for the automatic make-up gain and by callers who
        want the static curve); the per-sample curve runs on the device."""
        t, r, k = self._threshold, self._ratio, self._knee
        mode = self._mode
        if mode == DynamicsMode.LIMIT:
            r = float("inf")
        if mode in (DynamicsMode.COMPRESS, DynamicsMode.LIMIT):
            slope = -1.0 if math.isinf(r) else (1.0 / r - 1.0)
            if k <= 0:
                return (level_db - t) * slope if level_db > t else 0.0
            h = k / 2.0
            if level_db < t - h:
                return 0.0
            if level_db > t + h:
                return (level_db - t) * slope
            x = level_db - t + h
            return -(x ** 2) / (2 * k) if math.isinf(r) else slope * (x ** 2) / (2 * k)
        if mode == DynamicsMode.EXPAND:
            if k <= 0:
                return -(t - level_db) * (r - 1.0) if level_db < t else 0.0
            h = k / 2.0
            if level_db > t + h:
                return 0.0
            if level_db < t - h:
                return -(t - level_db) * (r - 1.0)
            x = t + h - level_db
            return -(r - 1.0) * (x ** 2) / (2 * k)
        if k <= 0:
            return self._range if level_db < t else 0.0
        h = k / 2.0
        if level_db > t + h:
            return 0.0
        if level_db < t - h:
            return self._range
        return (t + h - level_db) / k * self._range

    def _compute_auto_makeup(self) -> float:           # dynamics_pe.py:88-110
        if self._mode in (DynamicsMode.EXPAND, DynamicsMode.GATE):
            return 0.0
        return -self._compute_gain_db(self._threshold + 12.0) * 0.7

    # ------------------------------------------------------------------ device parameters
    def _param_record(self) -> dict:
        t, r, k = self._threshold, self._ratio, self._knee
        limit = self._mode == DynamicsMode.LIMIT or (self._mode == DynamicsMode.COMPRESS and math.isinf(r))
        mode = {DynamicsMode.COMPRESS: 0, DynamicsMode.LIMIT: 1, DynamicsMode.EXPAND: 2, DynamicsMode.GATE: 3}[self._mode]
        if limit:
            mode = 1
        slope = (1.0 / r - 1.0) if mode == 0 else (r - 1.0)
        if math.isinf(slope):
            slope = 0.0                                  # unused by the limit curve
        return dict(mode=mode, soft=1 if k > 0 else 0, stereo_link=1 if self._stereo_link else 0,
                    wide_makeup=1 if self._wide_makeup else 0,
                    threshold=t, slope=slope, neg_slope=-(r - 1.0) if not math.isinf(r) else 0.0,
                    half_knee=k / 2.0, two_knee=2 * k, knee=k, knee_lo=t - k / 2.0, knee_hi=t + k / 2.0,
                    gate_range=self._range, makeup=self._makeup_gain_db, gate_range_d=self._range,
                    makeup_d=self._makeup_gain_db)

    def _render(self, start: int, duration: int) -> Snippet:
        src = self._source.render(start, duration)
        env = self._envelope.render(start, duration)
        if self._params is None:
            self._params = _dev.upload_struct(_dev.DYNAMICS_PARAMS, **self._param_record())
        out = new_output(duration, src.channels)
        check(lib().pgx_dynamics(out.ptr, src.dev.ptr, env.dev.ptr, duration, src.channels, env.channels,
                                 self._params.ptr), "pgx_dynamics")
        return Snippet(start, out)

    def __repr__(self) -> str:
        makeup_str = "auto" if self._makeup_gain == self.AUTO else f"{self._makeup_gain_db:.1f}"
        return (f"DynamicsPE(threshold={self._threshold}, ratio={self._ratio}, knee={self._knee}, "
                f"makeup={makeup_str}, mode={self._mode.value}, stereo_link={self._stereo_link})")
