"""Host-side domain checks shared by GateSignal and TriggerSignal (gate_signal.py:63-99,
trigger_signal.py:70-121): shape / dtype of the rendered block, then a probe of its values -- every sample
with PYGMU_VALIDATE_SIGNALS_FULL=1, otherwise at most 64 evenly spaced ones including the first and the last.
Exception types and messages are the reference's; device-generated signals whose kernel guarantees the domain
skip all of this (it would cost a device-to-host copy per render)."""

from __future__ import annotations

import os

import numpy as np

PROBE_SAMPLES = 64


def env_flag(name: str, default: str = "0") -> bool:
    return os.environ.get(name, default).strip().lower() in ("1", "true", "yes", "on")


def probe_column(kind: str, block, full: bool) -> np.ndarray:
    """The values to inspect: channel 0 of a mono numeric (N, 1) block, thinned unless `full`."""
    if not isinstance(block, np.ndarray):
        raise TypeError(f"{kind} must render a numpy array, got {type(block)}")
    if block.ndim != 2 or block.shape[1] != 1:
        raise ValueError(f"{kind} must be mono with shape (N,1); got {block.shape}")
    if block.dtype.kind not in "fiu":
        raise TypeError(f"{kind} must render numeric dtype; got {block.dtype}")
    column = block[:, 0]
    if full or column.size <= PROBE_SAMPLES:
        return column
    return column[np.linspace(0, column.size - 1, num=PROBE_SAMPLES, dtype=int)]


def span(values) -> tuple:
    """(min, max) of the offending values, for the diagnostic."""
    return values.min(), values.max()
