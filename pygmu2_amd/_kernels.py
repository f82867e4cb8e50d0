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


def blitsaw_workspace(owner, batch: int, frames: int, streams: bool) -> DeviceBuffer | None:
    """Scratch for pgx_blitsaw's several-workgroups-per-oscillator form, cached on `owner` (grown on demand)."""
    need = lib().pgx_blitsaw_workspace_bytes(int(batch), int(frames), 1 if streams else 0)
    if not need:
        return None
    ws = getattr(owner, "_saw_workspace", None)
    if ws is None or ws.nbytes < need:
        ws = DeviceBuffer((need,), np.uint8)
        owner._saw_workspace = ws
    return ws


__all__ = ["lib", "new_output", "ptr", "check", "DeviceBuffer", "blitsaw_workspace"]
