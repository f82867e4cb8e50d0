"""
Read-ahead for small-block streaming of PURE sub-graphs.

A 1024-frame block is ~4 kB: at that size a render is nothing but launch latency and Python
overhead (the reference's own profile_biquad_vs_svfilter.py / AudioRenderer loops pull exactly
such blocks).  A pure PE is a function of (start, duration) only, so when a pure sub-graph is
pulled sequentially in small blocks the root of that sub-graph renders 64 blocks in ONE launch
sequence and hands out row-views of the resident result.  The samples are the same samples: every
kernel on these paths computes a frame from its absolute index alone (IdentityPE up to |index| < 2^24: beyond,
its numpy `arange` fill rule depends on the block start, and it declines the window).

Scope: SinePE with scalar parameters, GainPE, MixPE, ConstantPE, IdentityPE, DiracPE, ArrayPE, CropPE,
PeriodicGate, PeriodicTrigger -- and only when every input is itself eligible.  MixPE only while all
of its inputs have unbounded extents: its rule "skip an input whose extent misses the requested window"
(mix_pe.py:81-85) makes the output depend on the window, not just the frame index, as soon as an input
is non-zero outside its extent (ArrayPE / CropPE hold modes).
Disable with PYGMU_READ_AHEAD=0.
"""

from __future__ import annotations

import os
import threading

SMALL_BLOCK = 1 << 20       # requests up to this many frames are served from a resident window
FIRST_WINDOW_BLOCKS = 8     # the first refill of a stream; every further one is WINDOW_GROWTH times longer, up to
WINDOW_GROWTH = 8
AHEAD_BLOCKS = int(os.environ.get("PGX_READ_AHEAD_BLOCKS", "4096"))   # ... at most this many blocks per refill (8, 64, 512, 4096: with a
                            # served block at 0.1 - 0.2 us of host time -- csrc/_fast.c -- a refill per 64 blocks was half of
                            # what a 1024-frame block of C1 cost; a pure window costs HBM-rate rendering and its memory) ...
AHEAD_FRAMES = 1 << 25      # ... and about this many frames (128 MB per channel: a 44 100-frame pull refills 64
                            # blocks at a time, a 1 M-frame pull 32: launches of that size leave the ~4 us floor
                            # of a launch behind and stream at HBM rate)

_tls = threading.local()
_ENABLED = os.environ.get("PYGMU_READ_AHEAD", "1").strip().lower() not in ("0", "false", "no", "off")


class Declined(Exception):
    """Raised by a PE's _render while a window is being rendered for it (busy()): this particular range cannot be
    rendered in one piece with the blocks' samples (IdentityPE beyond 2^24).  The window is given up, the request takes
    the block-by-block path and the sub-graph stops opening windows."""


def busy() -> bool:
    return getattr(_tls, "busy", False)


def current_period() -> int:
    """Frames per caller block while a window is being rendered for a sub-graph that holds a PE whose samples depend on
    where a block begins (IdentityPE beyond 2^24), else 0.  Such a window is (first, end, buffer, period) and serves
    only the blocks it was cut for."""
    return getattr(_tls, "period", 0)


def _period_sensitive(pe) -> bool:
    """Does the sub-graph hold a period-sensitive PE, with nothing above it that changes (start, duration) on the way
    down?  (cached; a sensitive PE below something that re-addresses its pulls keeps declining such windows)"""
    cached = pe.__dict__.get("_ra_sensitive")
    if cached is None:
        def scan(node):
            own = bool(getattr(node, "_READ_AHEAD_PERIOD_SENSITIVE", False))
            below, ok = False, True
            for child in node.inputs():
                b, o = scan(child)
                below, ok = below or b, ok and o
            if below and not getattr(node, "_PASSES_BLOCKS", False):
                ok = False
            return own or below, ok
        has, ok = scan(pe)
        cached = pe.__dict__["_ra_sensitive"] = bool(has and ok)
    return cached


def enabled() -> bool:
    return _ENABLED


def set_enabled(flag: bool) -> None:
    global _ENABLED
    _ENABLED = bool(flag)


def eligible(pe) -> bool:
    """Pure, allow-listed, and all inputs eligible (cached on the instance; graphs are static)."""
    cached = pe.__dict__.get("_ra_ok")
    if cached is None:
        cached = bool(getattr(pe, "_READ_AHEAD_SAFE", False)) and pe.is_pure() and all(
            eligible(child) for child in pe.inputs())
        extra = getattr(pe, "_read_ahead_condition", None)     # PE-specific: e.g. MixPE's window-dependent skip rule
        if cached and extra is not None:
            cached = bool(extra())
        pe.__dict__["_ra_ok"] = cached
    return cached


def ahead_blocks(duration: int) -> int:
    return max(2, min(AHEAD_BLOCKS, AHEAD_FRAMES // max(1, duration)))


def render(pe, start: int, duration: int):
    """Serve (start, duration) from the PE's resident window, refilling it when the pull is
    sequential.  Returns None when the request should take the normal path.

    Instance state: `_ra_win` = (first frame, end frame, DeviceBuffer) of the resident window (device
    windows only; ProcessingElement.render serves hits from it directly), `_ra_last` = where the previous
    pull ended."""
    if not _ENABLED or duration > SMALL_BLOCK or getattr(_tls, "busy", False) or not eligible(pe):
        return None
    from .snippet import Snippet
    d = pe.__dict__
    win = d.get("_ra_win")
    if (win is not None and win[0] <= start and start + duration <= win[1]
            and (len(win) == 3 or (duration == win[3] and (start - win[0]) % win[3] == 0))):
        d["_ra_last"] = start + duration
        return Snippet.window_rows(start, win[2], start - win[0], duration)
    sequential = d.get("_ra_last") == start
    d["_ra_last"] = start + duration
    if not sequential:
        d["_ra_grow"] = FIRST_WINDOW_BLOCKS
        return None             # first pull / random access: render normally, remember where it ended
    grow = d.get("_ra_grow", FIRST_WINDOW_BLOCKS)      # slow start: 8, then 64 blocks (see look_ahead.py)
    d["_ra_grow"] = grow * WINDOW_GROWTH
    period = duration if _period_sensitive(pe) else 0
    _tls.busy = True
    _tls.period = period
    try:
        big = pe._render(start, duration * max(2, min(grow, ahead_blocks(duration))))
    except Declined:
        d["_ra_ok"] = False
        d.pop("_ra_win", None)
        return None
    finally:
        _tls.busy = False
        _tls.period = 0
    if not big.on_device:
        d.pop("_ra_win", None)
        return Snippet(start, big.data[:duration])
    d["_ra_win"] = (start, start + big.duration, big.dev, period) if period else (start, start + big.duration, big.dev)
    return Snippet.window_rows(start, big.dev, 0, duration)


def forget(pe) -> None:
    pe.__dict__.pop("_ra_win", None)
    pe.__dict__.pop("_ra_last", None)
