"""CropPE: limit a source to [start, start+duration) (crop_pe.py:54-96)."""

from __future__ import annotations

from .extent import ExtendMode, Extent
from .extent_window_pe import _ExtentWindowPE
from .processing_element import ProcessingElement


class CropPE(_ExtentWindowPE):
    def __init__(self, source: ProcessingElement, start: int, duration: int | None,
                 extend_mode: ExtendMode = ExtendMode.ZERO):
        if duration is not None and duration < 0:
            raise ValueError(f"duration must be >= 0, got {duration}")
        self._start = int(start)
        self._duration = None if duration is None else int(duration)
        end = None if self._duration is None else self._start + self._duration
        super().__init__(source, Extent(self._start, end), extend_mode)

    crop_extent = property(lambda self: self._extent)
    start = property(lambda self: self._start)
    duration = property(lambda self: self._duration)
    end = property(lambda self: self._extent.end)

    def __repr__(self) -> str:
        ext = f", extend_mode={self._extend_mode.value}" if self._extend_mode != ExtendMode.ZERO else ""
        return (f"CropPE(source={type(self._source).__name__}, start={self._start}, "
                f"end={self._extent.end}{ext})")
