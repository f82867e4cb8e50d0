"""
Extent: half-open sample-index interval [start, end) with optional infinite bounds.

Behavioural restatement of the reference's Extent / ExtendMode (extent.py:13-205):
None means unbounded on that side; start == end is the empty extent, which is falsy,
never intersects anything, and is the identity of union().
"""

from __future__ import annotations

from enum import Enum


class ExtendMode(Enum):
    ZERO = "zero"
    HOLD_FIRST = "hold_first"
    HOLD_LAST = "hold_last"
    HOLD_BOTH = "hold_both"


def _lo(a, b, pick):
    """Combine two lower bounds where None = -inf."""
    if a is None or b is None:
        return (b if a is None else a) if pick is max else None
    return pick(a, b)


def _hi(a, b, pick):
    """Combine two upper bounds where None = +inf."""
    if a is None or b is None:
        return (b if a is None else a) if pick is min else None
    return pick(a, b)


class Extent:
    __slots__ = ("_start", "_end")

    def __init__(self, start: int | None = None, end: int | None = None):
        if start is not None and end is not None and start > end:
            raise ValueError(f"start ({start}) must be less than or equal to end ({end})")
        self._start = start
        self._end = end

    start = property(lambda self: self._start)
    end = property(lambda self: self._end)

    @property
    def duration(self) -> int | None:
        if self._start is None or self._end is None:
            return None
        return self._end - self._start

    def is_empty(self) -> bool:
        return self._start is not None and self._start == self._end

    def contains(self, sample_index: int) -> bool:
        after_start = self._start is None or sample_index >= self._start
        before_end = self._end is None or sample_index < self._end
        return after_start and before_end

    def spans(self, start: int, duration: int) -> bool:
        if duration <= 0:
            return True
        if self._start is not None and start < self._start:
            return False
        return self._end is None or start + duration <= self._end

    def intersects(self, other: "Extent") -> bool:
        if self.is_empty() or other.is_empty():
            return False
        if self._end is not None and other._start is not None and self._end <= other._start:
            return False
        if other._end is not None and self._start is not None and other._end <= self._start:
            return False
        return True

    def intersection(self, other: "Extent") -> "Extent":
        for e in (self, other):
            if e.is_empty():
                return Extent(e._start, e._start)
        s = _lo(self._start, other._start, max)
        e = _hi(self._end, other._end, min)
        if s is not None and e is not None and s > e:
            return Extent(s, s)          # disjoint -> empty, anchored at the later start
        return Extent(s, e)

    def union(self, other: "Extent") -> "Extent":
        if self.is_empty():
            return other
        if other.is_empty():
            return self
        return Extent(_lo(self._start, other._start, min), _hi(self._end, other._end, max))

    def __eq__(self, other):
        if not isinstance(other, Extent):
            return NotImplemented
        return self._start == other._start and self._end == other._end

    def __hash__(self):
        return hash((self._start, self._end))

    def __bool__(self) -> bool:
        return not self.is_empty()

    def __repr__(self) -> str:
        s = "-∞" if self._start is None else str(self._start)
        e = "+∞" if self._end is None else str(self._end)
        return f"Extent({s}, {e})"
