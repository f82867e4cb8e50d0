"""_ExtentWindowPE: pass a source through inside a window, ExtendMode outside
(extent_window_pe.py:25-163).  Base of CropPE."""

from __future__ import annotations

from ._kernels import DeviceBuffer, check, lib, new_output
from .extent import ExtendMode, Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


class _ExtentWindowPE(ProcessingElement):
    _READ_AHEAD_SAFE = True

    def __init__(self, source: ProcessingElement, extent: Extent,
                 extend_mode: ExtendMode = ExtendMode.ZERO):
        self._source = source
        self._extent = extent
        self._extend_mode = extend_mode
        self._first_value: DeviceBuffer | None = None     # (1, C) device rows, fetched lazily
        self._last_value: DeviceBuffer | None = None

    source = property(lambda self: self._source)
    extent_window = property(lambda self: self._extent)
    extend_mode = property(lambda self: self._extend_mode)

    def inputs(self) -> list[ProcessingElement]:
        return [self._source]

    def is_pure(self) -> bool:
        return True

    def channel_count(self) -> int | None:
        return self._source.channel_count()

    def _compute_extent(self) -> Extent:
        return self._extent.intersection(self._source.extent())

    def _edge_value(self, which: str) -> DeviceBuffer | None:
        """Row of the source at the window's first / last sample (cached)."""
        if which == "first":
            if self._first_value is None and self._extent.start is not None:
                try:
                    self._first_value = self._source.render(self._extent.start, 1).dev
                except Exception:
                    return None
            return self._first_value
        if self._last_value is None and self._extent.end is not None and self._extent.end > 0:
            try:
                self._last_value = self._source.render(self._extent.end - 1, 1).dev
            except Exception:
                return None
        return self._last_value

    def _guess_channels(self) -> int:
        ch = self._source.channel_count()
        if ch is None:
            ins = self._source.inputs()
            if ins:
                ch = ins[0].channel_count()
        return 1 if ch is None else ch

    def _render(self, start: int, duration: int) -> Snippet:
        end = start + duration
        ws, we = self._extent.start, self._extent.end
        lo = start if ws is None else max(start, ws)
        hi = end if we is None else min(end, we)
        hold_first = self._extend_mode in (ExtendMode.HOLD_FIRST, ExtendMode.HOLD_BOTH)
        hold_last = self._extend_mode in (ExtendMode.HOLD_LAST, ExtendMode.HOLD_BOTH)
        L = lib()

        if lo == start and hi == end:
            return self._source.render(start, duration)          # fully inside the window: nothing to cut

        if lo >= hi:                                               # request misses the window
            ch = self._guess_channels()
            out = new_output(duration, ch)
            row = None
            if ws is not None and end <= ws and hold_first:
                row = self._edge_value("first")
            elif we is not None and start >= we and hold_last:
                row = self._edge_value("last")
            if row is None:
                check(L.pgx_fill(out.ptr, duration * ch, 0.0), "pgx_fill")
            else:
                # a one-row source with both holds on replicates the row everywhere
                check(L.pgx_window_copy(out.ptr, 0, duration, ch, row.ptr, 0, 1, 1, 1), "pgx_window_copy")
            return Snippet(start, out)

        seg = self._source.render(lo, hi - lo)
        ch = seg.channels
        out = new_output(duration, ch)
        # body: zero outside [lo, hi)
        check(L.pgx_window_copy(out.ptr, start, duration, ch, seg.dev.ptr, lo, hi - lo, 0, 0),
              "pgx_window_copy")
        if ws is not None and start < ws and hold_first:
            row = self._edge_value("first")
            if row is not None:
                check(L.pgx_window_copy(out.ptr, 0, ws - start, ch, row.ptr, 0, 1, 1, 1), "pgx_window_copy")
        if we is not None and end > we and hold_last:
            row = self._edge_value("last")
            if row is not None and we - start < duration:
                tail = DeviceBufferView(out, (we - start) * ch)
                check(L.pgx_window_copy(tail.ptr, 0, end - we, ch, row.ptr, 0, 1, 1, 1), "pgx_window_copy")
        return Snippet(start, out)


class DeviceBufferView:
    """Non-owning pointer into a DeviceBuffer at an element offset."""

    def __init__(self, buf: DeviceBuffer, element_offset: int):
        self.base = buf
        self.ptr = buf.offset_ptr(element_offset)
