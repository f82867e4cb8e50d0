#!/usr/bin/env python3
"""Launch pgx_biquad_const a few times at the bench sizes (for rocprofv3 --pmc passes).
argv[1] = "exact" selects the reduce + apply pair (settle_frames = 0)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import device
from pygmu2_amd.biquad_pe import rbj_coefficients, settle_frames

lib = device.ensure_init()
pg.set_sample_rate(44100)
for frames, reps in ((1_000_000, 5), (16_000_000, 5), (33_000_000, 5), (1 << 26, 3)):
    x = pg.SinePE(440.0).render(0, frames).dev
    out = device.DeviceBuffer((frames, 1), np.float32)
    c = rbj_coefficients(pg.BiquadMode.LOWPASS, 1000.0, 0.707, 0.0, 44100.0)
    coef = device.DeviceBuffer.from_host(np.asarray(c, dtype=np.float64))
    settle = 0 if sys.argv[1:] == ["exact"] else settle_frames(c[3], c[4])
    state = device.DeviceBuffer((1, 2), np.float64, zero=True)
    tables = device.DeviceBuffer((lib.pgx_biquad_table_doubles(),), np.float64)
    device.check(lib.pgx_biquad_tables(tables.ptr, coef.ptr, 1))
    ws = device.DeviceBuffer((max(lib.pgx_biquad_workspace_bytes(1, frames, 1, settle), 1),), np.uint8)
    for _ in range(reps):
        device.check(lib.pgx_biquad_const(out.ptr, 0, x.ptr, 0, 1, frames, 1, coef.ptr, tables.ptr if settle else None, settle, state.ptr,
                                          ws.ptr))
    device.synchronize()
print("done")
