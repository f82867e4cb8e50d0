#!/usr/bin/env python3
"""Launch pgx_biquad_const a few times at the bench sizes (for rocprofv3 --pmc passes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import device
from pygmu2_amd.biquad_pe import rbj_coefficients

lib = device.ensure_init()
pg.set_sample_rate(44100)
for frames, reps in ((1_000_000, 5), (1 << 26, 3)):
    x = pg.SinePE(440.0).render(0, frames).dev
    out = device.DeviceBuffer((frames, 1), np.float32)
    coef = device.DeviceBuffer.from_host(np.asarray(
        rbj_coefficients(pg.BiquadMode.LOWPASS, 1000.0, 0.707, 0.0, 44100.0), dtype=np.float64))
    state = device.DeviceBuffer((1, 2), np.float64, zero=True)
    ws = device.DeviceBuffer((max(lib.pgx_biquad_workspace_bytes(1, frames, 1), 1),), np.uint8)
    for _ in range(reps):
        device.check(lib.pgx_biquad_const(out.ptr, 0, x.ptr, 0, 1, frames, 1, coef.ptr, state.ptr, ws.ptr))
    device.synchronize()
print("done")
