"""
Light render diagnostics: per-PE pull counts and host-side timings, thread-local, off by
default.  Counterpart of the reference's diagnostics hooks that ProcessingElement.render
consults (processing_element.py:116-133).  Host timings include launch overhead only --
kernels run asynchronously; device time comes from rocprofv3 / HIP events (bench.py).
"""

from __future__ import annotations

import threading
from collections import defaultdict

_tls = threading.local()
_ACTIVE = 0          # threads with diagnostics on: ProcessingElement.render looks no further while this is 0


def _state():
    s = getattr(_tls, "s", None)
    if s is None:
        s = {"enabled": False, "pulls": False, "timing": False,
             "pull_counts": defaultdict(int), "timings": defaultdict(lambda: [0, 0])}
        _tls.s = s
    return s


def enable(pull_counts: bool = True, timing: bool = True) -> None:
    global _ACTIVE
    s = _state()
    if not s["enabled"]:
        _ACTIVE += 1
    s["enabled"], s["pulls"], s["timing"] = True, bool(pull_counts), bool(timing)


def disable() -> None:
    global _ACTIVE
    s = _state()
    if s["enabled"]:
        _ACTIVE -= 1
    s["enabled"] = False


def reset() -> None:
    s = _state()
    s["pull_counts"].clear()
    s["timings"].clear()


def is_enabled() -> bool:
    return _state()["enabled"]


def pull_count_enabled() -> bool:
    return _state()["pulls"]


def timing_enabled() -> bool:
    return _state()["timing"]


def record_pull(pe) -> None:
    _state()["pull_counts"][id(pe), type(pe).__name__] += 1


def record_timing(pe, ns: int) -> None:
    t = _state()["timings"][id(pe), type(pe).__name__]
    t[0] += 1
    t[1] += int(ns)


def report() -> dict:
    s = _state()
    return {"pulls": {k[1] + f"@{k[0]:x}": v for k, v in s["pull_counts"].items()},
            "timings_ns": {k[1] + f"@{k[0]:x}": tuple(v) for k, v in s["timings"].items()}}
