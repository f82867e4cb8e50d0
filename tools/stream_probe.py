#!/usr/bin/env python3
"""Streaming reference points on this GPU: float4-coalesced gain kernel (read+write) at 2^26 frames."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygmu2_amd import device
lib = device.ensure_init()
n = 1 << 26
x = device.DeviceBuffer((n,), np.float32, zero=True)
y = device.DeviceBuffer((n,), np.float32)
for name, fn in (("gain_const (rd+wr 8 B/elem)", lambda: lib.pgx_gain_const(y.ptr, x.ptr, n, 0.5)),
                 ("fill (wr 4 B/elem)", lambda: lib.pgx_fill(y.ptr, n, 1.0))):
    for _ in range(3): fn()
    e0, e1 = device.Event(), device.Event()
    e0.record()
    for _ in range(10): fn()
    e1.record()
    ms = e1.elapsed_ms_since(e0) / 10
    print(f"{name:32s} {ms*1e3:8.1f} us")
