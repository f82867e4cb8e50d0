"""Small helpers shared by the PE modules: library handle, output allocation, pointer
extraction.  Every PE `_render` goes through `lib()`; without the HIP library or a GPU
that raises RuntimeError (no CPU fallback)."""

from __future__ import annotations

import numpy as np

from . import device as _dev
from .device import DeviceBuffer, check


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _dev.ensure_init()
    return _LIB


def new_output(frames: int, channels: int, *, zero: bool = False) -> DeviceBuffer:
    return DeviceBuffer((int(frames), int(channels)), np.float32, zero=zero)


def ptr(buf) -> int | None:
    """Device address of a DeviceBuffer, or None (-> NULL) for an absent stream."""
    return None if buf is None else buf.ptr


__all__ = ["lib", "new_output", "ptr", "check", "DeviceBuffer"]
