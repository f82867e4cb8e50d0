"""
Look-ahead for small-block streaming of STATEFUL sub-graphs.

The reference's block loops (benchmarks/profile_biquad_vs_svfilter.py:75-87, audio_renderer.py:171-179,
utils.py:60-62) pull 1024-frame blocks.  A filter graph at that size is a chain of dependent launches, each
little more than its latency: six of them cost about 25 us per block however fast each kernel is.  The
filters, oscillators and envelopes here produce the same samples however a stream is cut into blocks (their
state is carried sample-exactly), so when a sub-graph is pulled sequentially in small blocks its top PE
renders up to `AHEAD_BLOCKS` blocks (about `AHEAD_FRAMES` frames) in one go and hands the caller row views
of that resident window -- the same samples, 1/64 of the launches.

What that must not change is what the caller can observe between blocks.  Before a window is rendered the
state of every stateful PE under it is snapshot (device blobs copied device-to-device, host-side fields by
value).  While the caller keeps pulling the next block, blocks are served from the window.  Anything else --
a seek, a pull of a PE inside the sub-graph, reset_state / on_start / on_stop anywhere in it -- first
`settle`s the window: the snapshot is restored and the part of the window the caller has actually consumed
is rendered again (output discarded), which leaves every state exactly where block-by-block rendering
would have left it; then the request takes the normal path.

A PE takes part when its class says so: `_LOOK_AHEAD_SAFE = True` promises block-partition invariance and
lists the mutable fields in `_STATE_FIELDS` (DeviceBuffers are copied, everything else is taken by value);
pure PEs qualify through read_ahead's allow-list.  `_look_ahead_condition()` may veto per instance.  One PE
that does not qualify anywhere below keeps the whole sub-graph on the block-by-block path.
Disable with PYGMU_LOOK_AHEAD=0.
"""

from __future__ import annotations

import os
import threading

import numpy as np

from . import read_ahead as _read_ahead

SMALL_BLOCK = 1 << 20       # pulls up to this many frames are served from a look-ahead window
FIRST_WINDOW_BLOCKS = 8     # the first window of a stream; every further one is WINDOW_GROWTH times longer, up to
WINDOW_GROWTH = 2           # (8, 16, 32, 64, 128, 256: a stream that stops after k blocks has rendered < 2k + 8; with x8 a
                            # 20-block stream rendered 8 + 33 blocks)
AHEAD_BLOCKS = int(os.environ.get("PGX_LOOK_AHEAD_BLOCKS", "256"))   # ... at most this many blocks per window (64 until round 4: the autowah
                            # graphs in 1024-frame blocks 519 / 544 Msamples/s at 64, 548 / 568 at 128, 581 / 592 at 256) ...
AHEAD_FRAMES = int(os.environ.get("PGX_LOOK_AHEAD_FRAMES", str(1 << 25)))      # ... and about this many frames (1 M-frame pulls: 32 blocks per window, 128 MB per
                            # channel of every PE in it: C2 3.8 us per step at 2^24, 3.5 at 2^25, 3.3 at 2^26)
AHEAD_FRAMES_SMALL_GRAPH = int(os.environ.get("PGX_LOOK_AHEAD_FRAMES_SMALL", str(1 << 27)))   # a window root with one or two PEs
                            # under it holds one or two such buffers, not dozens: 4x the frames (3 - 4 PEs: 2x).  C2 over a stream
                            # of 20 000 steps: 883 000 Msamples/s at 2^25, 978 000 at 2^26, 1 045 000 at 2^27 (fewer launch ramps
                            # and warm-up halves per frame, a quarter of the window openings)
_FRAMES_EXPLICIT = "PGX_LOOK_AHEAD_FRAMES" in os.environ


def frame_cap(n_nodes: int) -> int:
    """Frames per window for a window root with `n_nodes` PEs in its sub-graph (itself included)."""
    if _FRAMES_EXPLICIT or n_nodes > 4:
        return AHEAD_FRAMES
    return max(AHEAD_FRAMES, AHEAD_FRAMES_SMALL_GRAPH if n_nodes <= 2 else AHEAD_FRAMES_SMALL_GRAPH // 2)

STATS = {"window_frames": 0, "windows": 0}     # frames rendered into windows since the process started (bench.py reports
                                               # how many frames a timed region really rendered next to those it counts)
_tls = threading.local()
_ENABLED = os.environ.get("PYGMU_LOOK_AHEAD", "1").strip().lower() not in ("0", "false", "no", "off")


def enabled() -> bool:
    return _ENABLED


def set_enabled(flag: bool) -> None:
    global _ENABLED
    _ENABLED = bool(flag)


def _busy() -> bool:
    return getattr(_tls, "busy", False)


def current_period() -> int:
    """Frames per caller block while a window of a block-sensitive sub-graph is being rendered (else 0): PEs whose
    reference arithmetic restarts at block edges (EnvelopePE's RMS detector) cut their work there."""
    return getattr(_tls, "period", 0)


def _subtree(pe, seen, out):
    if id(pe) in seen:
        return
    seen.add(id(pe))
    out.append(pe)
    for child in pe.inputs():
        _subtree(child, seen, out)


def _node_ok(pe) -> bool:
    stateful_ok = bool(getattr(pe, "_LOOK_AHEAD_SAFE", False))
    pure_ok = bool(getattr(pe, "_READ_AHEAD_SAFE", False)) and pe.is_pure()
    if not (stateful_ok or pure_ok):
        return False
    if pure_ok and not stateful_ok:
        cond = getattr(pe, "_read_ahead_condition", None)
        if cond is not None and not cond():
            return False
    cond = getattr(pe, "_look_ahead_condition", None)
    return True if cond is None else bool(cond())


def capable(pe) -> bool:
    """The sub-graph under `pe` carries state and every PE in it qualifies (cached on the instance)."""
    cached = pe.__dict__.get("_la_ok")
    if cached is None:
        nodes = []
        _subtree(pe, set(), nodes)
        cached = (all(_node_ok(n) for n in nodes)
                  and any(getattr(n, "_STATE_FIELDS", None) is not None for n in nodes)
                  and not _read_ahead.eligible(pe))
        pe.__dict__["_la_ok"] = cached
        # block-sensitive PEs (arithmetic that restarts at the caller's block edges) are told the block length
        # while a window renders; that only works if everything above them hands (start, duration) down
        # unchanged, which a PE declares with `_PASSES_BLOCKS` -- otherwise the sub-graph stays block by block
        sens, seen = _sensitive_below(pe, {})
        if sens and not seen:
            cached = pe.__dict__["_la_ok"] = False
        pe.__dict__["_la_sensitive"] = bool(sens and cached)
    return cached


def _sensitive_below(pe, memo):
    """(does the sub-graph under pe hold a block-sensitive PE, do all PEs above those pass blocks through)."""
    if id(pe) in memo:
        return memo[id(pe)]
    f = getattr(pe, "_look_ahead_block_sensitive", None)
    own = f is not None and bool(f())
    any_below, ok = False, True
    for child in pe.inputs():
        s, o = _sensitive_below(child, memo)
        any_below = any_below or s
        ok = ok and o
    if any_below and not getattr(pe, "_PASSES_BLOCKS", False):
        ok = False
    memo[id(pe)] = (own or any_below, ok)
    return memo[id(pe)]


# ------------------------------------------------------------------------------------ snapshots
def _copy_value(value):
    from .device import DeviceBuffer, check, ensure_init
    if isinstance(value, DeviceBuffer):
        twin = DeviceBuffer(value.shape, value.dtype)
        if value.nbytes:
            check(ensure_init().pgx_memcpy_d2d(twin.ptr, value.ptr, value.nbytes), "pgx_memcpy_d2d")
        return twin
    if isinstance(value, np.ndarray):
        return value.copy()
    if isinstance(value, (list, dict, set)):
        return type(value)(value)
    return value                      # numbers, None, tuples, enums: immutable


def _snapshot_plan(nodes):
    """[(pe, state field names, pe._la_take_snapshot or None)] of the stateful PEs among `nodes` (graphs are static:
    made once per window root)."""
    plan = []
    for pe in nodes:
        fields = getattr(pe, "_STATE_FIELDS", None)
        if fields:
            plan.append((pe, fields, getattr(pe, "_la_take_snapshot", None)))     # a PE whose next kernel can write the copy itself
    return plan


def take_snapshot(nodes, plan=None):
    snap = []
    for pe, fields, own in (plan if plan is not None else _snapshot_plan(nodes)):
        saved = own() if own is not None else None
        if saved is None:
            saved = {name: _copy_value(getattr(pe, name)) for name in fields}
        snap.append((pe, saved))
    return snap


def restore_snapshot(snap) -> None:
    """Single use: the saved copies themselves become the PEs' state."""
    for pe, fields in snap:
        for name, value in fields.items():
            setattr(pe, name, value)


def _flush_pending_backups(snap) -> None:
    for n, _ in snap:
        if getattr(n, "_backup_target", None) is not None:     # (only a PE with the hook has the attribute)
            n._flush_backup()


def _expected_window_failure(exc) -> bool:
    """What a window `blocks` times longer than the caller's block may legitimately run into: a PE that declines it,
    the HBM for the intermediates, a kernel's size limit (PGX_ERR_INVALID -> ValueError)."""
    return isinstance(exc, (_read_ahead.Declined, MemoryError, ValueError))


_noted: set = set()


def _note_window_failure(pe, exc) -> None:
    key = (type(pe).__name__, type(exc).__name__, str(exc)[:80])
    if key not in _noted:                             # once per kind: a stream would repeat it block after block
        _noted.add(key)
        import logging
        logging.getLogger("pygmu2_amd.look_ahead").warning(
            "%s: look-ahead window declined (%s: %s); this PE renders block by block from here on",
            type(pe).__name__, type(exc).__name__, exc)


# ------------------------------------------------------------------------------------ windows
class _Window:
    __slots__ = ("first", "end", "buf", "served", "snap", "nodes", "block")     # block: 0, or the only block size served


def render(pe, start: int, duration: int):
    """Serve (start, duration) from the PE's window, or open one when the pull continues the previous
    one.  None: the request takes the normal path (any window has been settled by then)."""
    if _busy():
        return None
    d = pe.__dict__
    owner = d.get("_la_owner")
    if owner is not None:                             # pulled directly while inside somebody's window
        settle(owner)
    win = d.get("_la_win")
    if not _ENABLED:                                  # switched off mid-stream: open windows are settled, none opened
        if win is not None:
            settle(pe)
        return None
    if win is not None:
        if start == win.served and start + duration <= win.end and (not win.block or duration == win.block):
            win.served = start + duration
            d["_la_last"] = win.served                # the pull after the window's last block continues the stream
            from .snippet import Snippet
            return Snippet.window_rows(start, win.buf, start - win.first, duration)
        settle(pe)
    if duration > SMALL_BLOCK or not capable(pe):
        return None
    sequential = d.get("_la_last") == start
    d["_la_last"] = start + duration
    if not sequential:
        d["_la_grow"] = FIRST_WINDOW_BLOCKS           # a new stream: start small again
        return None
    from .snippet import Snippet
    nodes = d.get("_la_nodes")                        # graphs are static: the walk is done once
    if nodes is None:
        nodes = []
        _subtree(pe, set(), nodes)
        d["_la_nodes"] = nodes
        d["_la_plan"] = _snapshot_plan(nodes)
        d["_la_below"] = [n for n in nodes if n is not pe]
    for n in d["_la_below"]:                          # a window below (opened while this PE was pulled one
        if "_la_win" in n.__dict__:                   # level down) is closed first
            settle(n)
    snap = take_snapshot(nodes, d["_la_plan"])
    block = duration if d.get("_la_sensitive") else 0
    _tls.busy = True
    _tls.period = block
    # slow start: a stream that stops after a few blocks has not paid for 64; one that keeps going doubles its
    # window with every refill (8, 16, 32, 64 blocks)
    grow = d.get("_la_grow", FIRST_WINDOW_BLOCKS)
    blocks = max(2, min(grow, AHEAD_BLOCKS, frame_cap(len(nodes)) // duration))
    try:
        big = pe._render(start, duration * blocks)
    except BaseException as exc:
        # a render `blocks` times longer can fail where the block itself would not (HBM for the intermediates,
        # a kernel's size limit, a PE that declines the window): every state goes back to the snapshot, this PE stops
        # opening windows and the request takes the block-by-block path -- the caller sees what it would have seen
        # without look-ahead.  Anything else (a HIP runtime error, a bug) is not swallowed: the states are put back
        # and the exception goes to the caller, so that a fault is diagnosed where it first shows.
        _tls.busy = False
        _tls.period = 0
        d["_la_ok"] = False
        try:
            _flush_pending_backups(snap)
            restore_snapshot(snap)
        except Exception:                             # noqa: BLE001 -- the device is gone: the first error is the story
            if _expected_window_failure(exc):
                raise
        if not _expected_window_failure(exc):
            raise
        _note_window_failure(pe, exc)
        return None
    finally:
        _tls.busy = False
        _tls.period = 0
    # a snapshot copy that the window's own kernel was to write (BiquadPE._la_take_snapshot) and that no render asked
    # for -- CropPE past its end returns fill without pulling its source -- is made now: the state is unchanged when
    # the PE was never rendered, so the copy is the state before the window
    _flush_pending_backups(snap)
    d["_la_grow"] = grow * WINDOW_GROWTH
    STATS["window_frames"] += duration * blocks
    STATS["windows"] += 1
    if not big.on_device:                             # host-side graph: nothing to gain, nothing was assumed
        restore_snapshot(snap)
        _tls.busy = True
        try:
            return pe._render(start, duration)
        finally:
            _tls.busy = False
    win = _Window()
    win.first, win.end, win.buf = start, start + big.duration, big.dev
    win.served, win.snap, win.nodes, win.block = start + duration, snap, nodes, block
    d["_la_win"] = win
    for n in d["_la_below"]:
        n.__dict__["_la_owner"] = pe
    return Snippet.window_rows(start, win.buf, 0, duration)


def settle(owner) -> None:
    """Close `owner`'s window: every state under it goes to where the consumed part of the window ends."""
    win = owner.__dict__.pop("_la_win", None)
    if win is None:
        return
    for n in win.nodes:
        if n.__dict__.get("_la_owner") is owner:
            del n.__dict__["_la_owner"]
    if win.served >= win.end:
        return                                        # consumed to the last frame: the states are already there
    restore_snapshot(win.snap)
    if win.served > win.first:
        was = _busy()
        _tls.busy = True
        _tls.period = win.block
        try:
            owner._render(win.first, win.served - win.first)
        finally:
            _tls.busy = was
            _tls.period = 0


def before_direct_access(pe) -> None:
    """Called (outside a look-ahead render) when a PE inside somebody's window is pulled or reset itself."""
    if _busy():
        return
    owner = pe.__dict__.get("_la_owner")
    if owner is not None:
        settle(owner)
    if pe.__dict__.get("_la_win") is not None:
        settle(pe)


def forget(pe) -> None:
    pe.__dict__.pop("_la_last", None)
