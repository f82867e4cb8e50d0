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


def _as_transition(kind) -> TransitionType:
    """A TransitionType, or its name in any case; an unknown name means LINEAR (piecewise_pe.py:124-130)."""
    if isinstance(kind, TransitionType):
        return kind
    by_name = {t.value: t for t in TransitionType}
    return by_name.get(str(kind).lower(), TransitionType.LINEAR)


def _break_points(points):
    """Break points as two device-ready columns ordered by time: int64 sample indices (fractions truncate, as a
    float64 -> int64 cast does) and float64 values.  Points that share a time keep the order numpy's default argsort
    gives them -- the order the reference's curve sees (piecewise_pe.py:31-43)."""
    table = np.asarray(list(points), dtype=np.float64).reshape(-1, 2)
    when = table[:, 0].astype(np.int64)
    rank = np.argsort(when)
    return np.ascontiguousarray(when[rank]), np.ascontiguousarray(table[rank, 1])


class PiecewisePE(SourcePE):
    _READ_AHEAD_SAFE = True

    def __init__(self, points: Sequence[Tuple[int, float]],
                 transition_type: TransitionType | str = TransitionType.LINEAR,
                 extend_mode: ExtendMode = ExtendMode.ZERO, channels: int = 1):
        if len(points) == 0:
            raise ValueError("PiecewisePE requires at least one point")
        if int(channels) < 1:
            raise ValueError(f"channels must be >= 1, got {int(channels)}")
        self._times, self._values = _break_points(points)
        self._n = int(self._times.shape[0])
        self._transition_type = _as_transition(transition_type)
        self._extend_mode = extend_mode
        self._channels = int(channels)
        self._times_dev: DeviceBuffer | None = None      # uploaded on the first render
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
