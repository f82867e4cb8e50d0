#!/usr/bin/env python3
"""Launch pgx_biquad_sine (BiquadPE(SinePE) as one kernel) a few times at the bench sizes (rocprofv3 --pmc passes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import device
from pygmu2_amd.biquad_pe import rbj_coefficients, settle_frames
lib = device.ensure_init()
c = rbj_coefficients(pg.BiquadMode.LOWPASS, 1000.0, 0.707, 0.0, 44100.0)
coef = device.DeviceBuffer.from_host(np.asarray(c, dtype=np.float64))
settle = settle_frames(c[3], c[4])
state = device.DeviceBuffer((1, 2), np.float64, zero=True)
tables = device.DeviceBuffer((lib.pgx_biquad_table_doubles(),), np.float64)
device.check(lib.pgx_biquad_tables(tables.ptr, coef.ptr, 1))
w = 2.0 * np.pi * 440.0
for frames, reps in ((1_000_000, 5), (16_000_000, 5), (33_000_000, 5), (1 << 26, 3), (134_000_000, 3)):
    out = device.DeviceBuffer((frames, 1), np.float32)
    for _ in range(reps):
        device.check(lib.pgx_biquad_sine(out.ptr, 10 ** 9, frames, 44100.0, w, 1.0, 0.0, coef.ptr, tables.ptr, settle, state.ptr, None))
    device.synchronize()
print("done")
