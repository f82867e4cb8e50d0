#!/usr/bin/env python3
"""Launch pgx_convolve (C3: 96 000 stereo frames x 65 536 taps) a few times (for rocprofv3 passes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygmu2_amd import device
lib = device.ensure_init()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 96_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
x = (np.random.default_rng(0).standard_normal((frames, 2)) * 0.1).astype(np.float32)
h = (np.random.default_rng(1).standard_normal(65536) * np.exp(-np.arange(65536) / 8000.0)).astype(np.float32)
xd, hd = device.DeviceBuffer.from_host(x), device.DeviceBuffer.from_host(h.reshape(-1, 1))
out = device.DeviceBuffer((frames, 2), np.float32)
hist = device.DeviceBuffer((65535, 2), np.float32, zero=True)
ws = device.DeviceBuffer((lib.pgx_convolve_workspace_bytes(frames, 65536, 2),), np.uint8)
e0, e1 = device.Event(), device.Event()
for i in range(reps + 2):
    if i == 2:
        e0.record()
    device.check(lib.pgx_convolve(out.ptr, xd.ptr, frames, 2, hd.ptr, 65536, 1, 2, hist.ptr, ws.ptr))
e1.record()
ms = e1.elapsed_ms_since(e0) / reps
print(f"{frames} frames: {ms*1e3:.1f} us/launch  {2*65536*2*frames/ms/1e9:.1f} TFLOP/s  {frames/ms/1e3:.1f} Msamples/s")
