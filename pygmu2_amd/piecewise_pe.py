"""
PiecewisePE: (sample_index, value) break-point curve with step / linear / exponential /
sigmoid / constant-power transitions (piecewise_pe.py:20-244).  Break points live in HBM;
every output frame finds its segment by binary search (pgx_piecewise).
"""

from __future__ import annotations

from enum import Enum
from typing import List, Sequence, Tuple

import numpy as np

from ._kernels import DeviceBuffer, check, lib, new_output
from .extent import ExtendMode, Extent
from .snippet import Snippet
from .source_pe import SourcePE


class TransitionType(Enum):
    STEP = "step"
    LINEAR = "linear"
    EXPONENTIAL = "exponential"
    SIGMOID = "sigmoid"
    CONSTANT_POWER = "constant_power"


_TRANSITION_INDEX = {t: i for i, t in enumerate(TransitionType)}


class PiecewisePE(SourcePE):
    _READ_AHEAD_SAFE = True

    def __init__(self, points: Sequence[Tuple[int, float]],
                 transition_type: TransitionType | str = TransitionType.LINEAR,
                 extend_mode: ExtendMode = ExtendMode.ZERO, channels: int = 1):
        if not points:
            raise ValueError("PiecewisePE requires at least one point")
        arr = np.array(points, dtype=np.float64)
        times = arr[:, 0].astype(np.int64)
        values = arr[:, 1].astype(np.float64)
        order = np.argsort(times)                 # same (default) sort as the reference, piecewise_pe.py:38
        self._times, self._values = times[order], values[order]
        self._n = len(self._times)
        if isinstance(transition_type, str):
            try:
                transition_type = TransitionType(transition_type.lower())
            except ValueError:
                transition_type = TransitionType.LINEAR
        self._transition_type = transition_type
        self._extend_mode = extend_mode
        self._channels = int(channels)
        if self._channels < 1:
            raise ValueError(f"channels must be >= 1, got {self._channels}")
        self._times_dev: DeviceBuffer | None = None
        self._values_dev: DeviceBuffer | None = None

    @property
    def points(self) -> List[Tuple[int, float]]:
        return list(zip(self._times.tolist(), self._values.tolist()))

    transition_type = property(lambda self: self._transition_type)
    extend_mode = property(lambda self: self._extend_mode)

    def _compute_extent(self) -> Extent:
        if self._extend_mode != ExtendMode.ZERO:
            return Extent(None, None)
        t0, t_last = int(self._times[0]), int(self._times[-1])
        return Extent(t0, t0 + 1) if self._n == 1 else Extent(t0, t_last)

    def channel_count(self) -> int:
        return self._channels

    def _render(self, start: int, duration: int) -> Snippet:
        if self._times_dev is None:
            self._times_dev = DeviceBuffer.from_host(self._times)
            self._values_dev = DeviceBuffer.from_host(self._values)
        out = new_output(duration, self._channels)
        hold_first = self._extend_mode in (ExtendMode.HOLD_FIRST, ExtendMode.HOLD_BOTH)
        hold_last = self._extend_mode in (ExtendMode.HOLD_LAST, ExtendMode.HOLD_BOTH)
        check(lib().pgx_piecewise(out.ptr, start, duration, self._channels, self._times_dev.ptr,
                                  self._values_dev.ptr, self._n, _TRANSITION_INDEX[self._transition_type],
                                  int(hold_first), int(hold_last)), "pgx_piecewise")
        return Snippet(start, out)

    def __repr__(self) -> str:
        return (f"PiecewisePE(points={self.points!r}, transition_type={self._transition_type.value}, "
                f"extend_mode={self._extend_mode.value}, channels={self._channels})")
