"""
ProcessingElement: the pull-model contract every PE implements.

Restates the reference's abstract base (processing_element.py:28-363): `render(start,
duration)` always yields exactly `duration` frames (zeros outside `extent()`), rejects
negative durations, short-circuits zero-length requests, and dispatches to the
subclass's `_render`.  Construction requires the global sample rate.  Lifecycle hooks
(`on_start`, `on_stop`, `reset_state`) call `_on_start` / `_on_stop` / `_reset_state`
when a subclass defines them.

Differences that come from running on the device: parameter streams are handed to the
kernels as device pointers (`_control_stream`) instead of float64 numpy vectors; the
float32 -> float64 widening the reference does on the host happens inside the kernels.
"""

from __future__ import annotations

import time
from abc import ABC, abstractmethod

import numpy as np

from . import device as _dev
from . import diagnostics as _diag
from . import look_ahead as _look_ahead
from . import read_ahead as _read_ahead
from .config import get_sample_rate, handle_error
from .diagnostics import is_enabled, pull_count_enabled, record_pull, record_timing, timing_enabled
from .extent import Extent
from .snippet import Snippet


_RA_LIMIT = _read_ahead.SMALL_BLOCK
_LA_LIMIT = _look_ahead.SMALL_BLOCK


class ProcessingElement(ABC):
    _sample_rate: int | None = None
    _cached_extent: Extent | None = None
    _READ_AHEAD_SAFE = False      # see read_ahead.py: set by PEs whose frames depend on the index alone

    def __new__(cls, *args, **kwargs):
        rate = get_sample_rate()
        if rate is None:
            raise RuntimeError(
                "Global sample_rate is required but not set. "
                "Call pygmu2_amd.set_sample_rate(rate) before constructing PEs.")
        obj = super().__new__(cls)
        obj._sample_rate = rate
        return obj

    # ------------------------------------------------------------------ identity
    @property
    def sample_rate(self) -> int | None:
        if self._sample_rate is not None:
            return self._sample_rate
        found = None
        for pe in self.inputs():
            r = pe.sample_rate
            if r is None:
                continue
            if found is None:
                found = r
            elif found != r:
                handle_error(f"{type(self).__name__}.sample_rate inferred conflicting input rates: "
                             f"{found} vs {r}. Using {found}.", fatal=False)
                break
        return found

    # ------------------------------------------------------------------ rendering
    def render(self, start: int, duration: int) -> Snippet:
        if duration <= 0:
            if duration < 0:
                raise ValueError(f"duration must be >= 0, got {duration}")
        elif not _diag._ACTIVE:
            # the two hot exits first: the next block of a stream served from a resident window
            d = self.__dict__
            win = d.get("_la_win")                       # look-ahead window of a stateful sub-graph
            if win is not None:
                end = start + duration
                if start == win.served and end <= win.end and (not win.block or duration == win.block):
                    win.served = d["_la_last"] = end
                    return Snippet.window_rows(start, win.buf, start - win.first, duration)
            else:
                win = d.get("_ra_win")                   # read-ahead window of a pure sub-graph: (first, end, buffer[, period])
                if (win is not None and win[0] <= start and start + duration <= win[1]
                        and (len(win) == 3 or (duration == win[3] and (start - win[0]) % win[3] == 0))):
                    d["_ra_last"] = start + duration
                    return Snippet.window_rows(start, win[2], start - win[0], duration)
        diag = is_enabled() if _diag._ACTIVE else False
        if diag and pull_count_enabled():
            record_pull(self)
        if duration == 0:
            ch = self.channel_count()
            return Snippet.from_zeros(start, 0, int(ch) if ch is not None else 1)
        if diag and timing_enabled():
            if "_la_win" in self.__dict__ or "_la_owner" in self.__dict__:
                _look_ahead.before_direct_access(self)   # timing turned on mid-stream: an open window is settled first
            t0 = time.perf_counter_ns()
            out = self._render(start, duration)
            record_timing(self, time.perf_counter_ns() - t0)
            return out
        # pure sub-graphs, sequential small pulls (the verdict of read_ahead.eligible is cached on the instance:
        # a PE that is not eligible skips the call altogether)
        if duration <= _RA_LIMIT:
            d = self.__dict__
            win = d.get("_ra_win")
            if (win is not None and win[0] <= start and start + duration <= win[1]
                    and (len(win) == 3 or (duration == win[3] and (start - win[0]) % win[3] == 0))):
                d["_ra_last"] = start + duration
                return Snippet.window_rows(start, win[2], start - win[0], duration)
            if d.get("_ra_ok", True):
                ahead = _read_ahead.render(self, start, duration)
                if ahead is not None:
                    return ahead
            # stateful sub-graphs pulled in small sequential blocks (look_ahead.py); a PE inside somebody's
            # window, or holding one, always goes through it so that the window is settled first
            if (duration <= _LA_LIMIT and d.get("_la_ok", True)) or "_la_win" in d or "_la_owner" in d:
                ahead = _look_ahead.render(self, start, duration)
                if ahead is not None:
                    return ahead
        elif "_la_win" in self.__dict__ or "_la_owner" in self.__dict__:
            _look_ahead.before_direct_access(self)
        return self._render(start, duration)

    @abstractmethod
    def _render(self, start: int, duration: int) -> Snippet:
        ...

    @abstractmethod
    def inputs(self) -> list["ProcessingElement"]:
        ...

    # ------------------------------------------------------------------ static properties
    def extent(self) -> Extent:
        if self._cached_extent is None:
            self._cached_extent = self._compute_extent()
        return self._cached_extent

    def _compute_extent(self) -> Extent:
        return Extent(None, None)

    def is_pure(self) -> bool:
        return False

    def channel_count(self) -> int | None:
        return None

    def required_input_channels(self) -> int | None:
        return None

    def resolve_channel_count(self, input_channel_counts: list[int]) -> int:
        if input_channel_counts:
            return input_channel_counts[0]
        raise ValueError(f"{type(self).__name__} has no inputs but channel_count() is None")

    # ------------------------------------------------------------------ lifecycle
    def on_start(self) -> None:
        _look_ahead.before_direct_access(self)
        _look_ahead.forget(self)
        _read_ahead.forget(self)
        hook = getattr(self, "_on_start", None)
        if hook is not None:
            hook()

    def on_stop(self) -> None:
        _look_ahead.before_direct_access(self)
        _look_ahead.forget(self)
        _read_ahead.forget(self)
        hook = getattr(self, "_on_stop", None)
        if hook is not None:
            hook()

    def reset_state(self) -> None:
        _look_ahead.before_direct_access(self)
        hook = getattr(self, "_reset_state", None)
        if hook is not None:
            hook()

    # ------------------------------------------------------------------ parameter helpers
    def _control_stream(self, param, start: int, duration: int, *, channel: int = 0):
        """
        Device-side counterpart of the reference's `_scalar_or_pe_values`
        (processing_element.py:296-363) for 1-D control parameters.

        Returns (scalar, stream): for a scalar parameter (float(param), None); for a PE
        parameter (None, DeviceBuffer of shape (duration, 1)) holding the selected channel
        of the rendered parameter.  The keep-alive of the rendered Snippet is the returned
        buffer itself.
        """
        if isinstance(param, ProcessingElement):
            snip = param.render(start, duration)
            buf = snip.dev
            ch = snip.channels
            if ch < 1:
                raise ValueError(f"param PE returned invalid shape {(snip.duration, ch)}")
            if channel < 0 or channel >= ch:
                raise ValueError(f"channel {channel} out of range for param with {ch} channels")
            if ch == 1:
                return None, buf
            mono = _dev.DeviceBuffer((duration, 1), np.float32)
            _dev.check(_dev.ensure_init().pgx_extract_channel(mono.ptr, buf.ptr, duration, ch, channel),
                       "pgx_extract_channel")
            return None, mono
        return float(param), None

    def _scalar_or_pe_values(self, param, start: int, duration: int, *, dtype=None, channel: int = 0,
                             allow_multichannel: bool = False, channels: int | None = None):
        """Host-side variant with the reference's exact signature and return values
        (numpy arrays); used by host-only PEs and by tests of the contract."""
        if dtype is None:
            dtype = np.float64
        if duration <= 0:
            if allow_multichannel:
                return np.zeros((0, channels if channels is not None else 1), dtype=dtype)
            return np.zeros((0,), dtype=dtype)
        if isinstance(param, ProcessingElement):
            data = param.render(start, duration).data
            if allow_multichannel:
                return data.astype(dtype, copy=False)
            if data.ndim != 2 or data.shape[1] < 1:
                raise ValueError(f"param PE returned invalid shape {getattr(data, 'shape', None)}")
            if channel < 0 or channel >= data.shape[1]:
                raise ValueError(f"channel {channel} out of range for param with {data.shape[1]} channels")
            return data[:, channel].astype(dtype, copy=False)
        value = float(param)
        if allow_multichannel:
            return np.full((duration, channels if channels is not None else 1), value, dtype=dtype)
        return np.full((duration,), value, dtype=dtype)


# ---------------------------------------------------------------------------------------------- hot paths in C
def _install_fast_paths() -> bool:
    """ProcessingElement.render's two hot exits (the next block of a stream served from a resident window) and
    Snippet.__del__ as C method descriptors (csrc/_fast.c, built by pygmu2_amd.build): the same semantics at a third of
    the interpreter time per small block.  Everything else still runs the Python functions above, and so does every
    pull when the module has not been built (PYGMU_FAST=0 switches it off)."""
    import os
    if os.environ.get("PYGMU_FAST", "1").strip().lower() in ("0", "false", "no", "off"):
        return False
    try:
        from . import _fast
    except ImportError:
        return False
    from . import snippet as _snippet
    _fast.install(ProcessingElement, Snippet, _look_ahead._Window, ProcessingElement.__dict__["render"],
                  _snippet.Snippet.__dict__["__del__"], vars(_diag))
    return True


FAST_PATHS = _install_fast_paths()
