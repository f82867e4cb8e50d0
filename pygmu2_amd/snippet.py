"""
Snippet: `start` + a (frames, channels) float32 payload.

Same contract as the reference's Snippet (snippet.py:26-109) -- 1-D input becomes
(N, 1), every dtype is normalised to float32 -- but the payload normally lives in
MI355X HBM (a DeviceBuffer produced by a HIP kernel).  `.data` is the reference's
attribute: it returns a numpy view of the payload, copied device->host on first use and
cached.  `.dev` returns the DeviceBuffer (uploading a host-constructed payload on first
use), which is what downstream PEs consume, so a chain of PEs never leaves the device.

A payload may still be in flight on another stream (the RCCL all-reduce of a sharded mix):
`ready` is then a callable that orders the library stream behind that work; it runs once, the
first time the payload is touched or when the Snippet is dropped (so the buffer cannot return to
the pool early).
"""

from __future__ import annotations

import numpy as np

from .device import DeviceBuffer


class Snippet:
    __slots__ = ("_start", "_host", "_dev", "_shape", "_ready", "_copy", "_base", "_bank_window")

    def __init__(self, start: int, data, ready=None):
        self._start = int(start)
        self._ready = ready
        self._copy = None
        self._base = None
        self._bank_window = False     # a row of a voice bank's mixed window (sharding.py reduces such a window whole)
        if isinstance(data, DeviceBuffer):
            if data.dtype != np.float32 or len(data.shape) != 2:
                raise ValueError("device payload must be float32 of shape (frames, channels)")
            self._dev = data
            self._host = None
            self._shape = data.shape
            return
        data = np.asarray(data)
        if data.ndim == 1:
            data = data.reshape(-1, 1)
        elif data.ndim != 2:
            raise ValueError(f"data must be 1D or 2D, got {data.ndim}D")
        if data.dtype != np.float32:
            data = data.astype(np.float32, copy=False)
        self._host = data
        self._dev = None
        self._shape = data.shape

    @classmethod
    def window_rows(cls, start: int, window: DeviceBuffer, first_row: int, rows: int) -> "Snippet":
        """`rows` frames of a resident (frames, channels) window, starting at `first_row`: what read-ahead hands
        out per small block.  The DeviceBuffer view is only built if somebody asks for `.dev` / `.data`."""
        self = object.__new__(cls)
        self._start = start
        self._ready = None
        self._copy = None
        self._host = None
        self._dev = None
        self._base = (window, first_row)
        self._bank_window = False
        self._shape = (rows, window.shape[1])
        return self

    @property
    def start(self) -> int:
        return self._start

    @property
    def end(self) -> int:
        return self._start + self._shape[0]

    @property
    def duration(self) -> int:
        return self._shape[0]

    @property
    def channels(self) -> int:
        return self._shape[1]

    def _resolve(self) -> None:
        ready, self._ready = self._ready, None
        if ready is not None:
            ready()

    def __del__(self):
        try:
            ready = self._ready
            if ready is not None and getattr(ready, "on_use_only", False):
                self._ready = None              # (a row of a window somebody else keeps alive and waits for)
            elif ready is not None and getattr(ready, "on_drop", None) is not None:
                self._ready = None              # (dropped unread: the producer has its own way of letting go)
                ready.on_drop()
            self._resolve()
            if self._copy is not None:          # a prefetch nobody read: the payload must outlive the copy
                from .device import fence_to_host
                fence_to_host(self._copy)
        except Exception:
            pass

    def prefetch(self) -> "Snippet":
        """Start moving the payload to the host (pinned block, the library's copy stream) without waiting:
        `.data` will block only for what is left of the copy.  For callers that read every block
        (benchmark_pes.py:176-185): render block k+1, then read block k."""
        if self._host is None and self._copy is None and self._shape[0]:
            self._host, self._copy = self.dev.begin_to_host()
        return self

    @property
    def data(self) -> np.ndarray:
        """Host view (frames, channels) float32; treat as immutable."""
        if self._copy is None:
            if self._host is not None:
                return self._host
            self.prefetch()
            if self._copy is None:              # zero frames
                self._host = np.zeros(self._shape, dtype=np.float32)
                return self._host
        from .device import wait_to_host
        ticket, self._copy = self._copy, None
        wait_to_host(ticket)
        return self._host

    @property
    def dev(self) -> DeviceBuffer:
        """Device payload (uploads a host-built snippet once)."""
        self._resolve()
        if self._dev is None:
            if self._base is not None:
                window, first_row = self._base
                self._dev = window.rows(first_row, self._shape[0])
            else:
                self._dev = DeviceBuffer.from_host(np.ascontiguousarray(self._host))
        return self._dev

    @property
    def on_device(self) -> bool:
        return self._dev is not None or self._base is not None

    @classmethod
    def from_zeros(cls, start: int, duration: int, channels: int = 1) -> "Snippet":
        return cls(start, np.zeros((duration, channels), dtype=np.float32))

    def __repr__(self) -> str:
        return f"Snippet(start={self._start}, duration={self.duration}, channels={self.channels})"

    def __eq__(self, other):
        if not isinstance(other, Snippet):
            return NotImplemented
        return (self._start == other._start and self._shape == other._shape
                and np.allclose(self.data, other.data))

    __hash__ = None
