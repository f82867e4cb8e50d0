"""IdentityPE: sample value == sample index (identity_pe.py:40-60)."""

from __future__ import annotations

import numpy as np

from ._kernels import check, lib, new_output
from .snippet import Snippet
from .source_pe import SourcePE


class IdentityPE(SourcePE):
    def __init__(self, channels: int = 1):
        self._channels = channels

    # read_ahead.py / look_ahead.py: below 2^24 every index is a float32, whatever block it is rendered in; beyond, the
    # fill depends on the block's start: a read-ahead window of equal blocks is filled block by block (pgx_ramp_blocks)
    _READ_AHEAD_SAFE = True
    _READ_AHEAD_PERIOD_SENSITIVE = True

    def _render(self, start: int, duration: int) -> Snippet:
        from . import look_ahead, read_ahead
        if read_ahead.busy() and not look_ahead._busy():
            period = read_ahead.current_period()
            if period and duration % period == 0:
                out = new_output(duration, self._channels)
                check(lib().pgx_ramp_blocks(out.ptr, start, duration, self._channels, period), "pgx_ramp_blocks")
                return Snippet(start, out)
        if max(abs(start), abs(start + duration)) >= 1 << 24:
            if read_ahead.busy() or look_ahead._busy():
                raise read_ahead.Declined("IdentityPE beyond 2^24: the arange fill depends on the block start")
        out = new_output(duration, self._channels)
        # The reference builds np.arange(start, start+n, dtype=float32); numpy fills that as
        # first + i*delta in float32 with first = float32(start), delta = float32(start+1) - first.
        first = np.float32(start)
        delta = np.float32(start + 1) - first
        check(lib().pgx_ramp(out.ptr, float(first), float(delta), duration, self._channels), "pgx_ramp")
        return Snippet(start, out)

    def channel_count(self) -> int:
        return self._channels

    def __repr__(self) -> str:
        return f"IdentityPE(channels={self._channels})"
