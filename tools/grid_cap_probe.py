#!/usr/bin/env python3
"""HBM-bound elementwise kernels at look-ahead window sizes under different grid caps (PGX_GRID_CAP workgroups per CU):
fill (ConstantPE), gain (8 B/frame), mix of two (12 B/frame), HIP events over 100 launches after 20 (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import pygmu2_amd as pg
from pygmu2_amd import device
pg.set_sample_rate(44100)
lib = device.ensure_init()
res = []
for frames in (11_289_600, 134_000_000):
    a = device.DeviceBuffer((frames, 1), np.float32, zero=True)
    b = device.DeviceBuffer((frames, 1), np.float32, zero=True)
    o = device.DeviceBuffer((frames, 1), np.float32)
    t = bench.event_avg_ms(lambda: device.check(lib.pgx_fill(o.ptr, frames, 0.25), "fill"), 100, 20)
    res.append(f"fill {frames}: {t * 1e3:7.2f} us {4 * frames / t / 1e9:5.2f} TB/s")
    t = bench.event_avg_ms(lambda: device.check(lib.pgx_gain_const(o.ptr, a.ptr, frames, 0.5), "gain"), 100, 20)
    res.append(f"gain {frames}: {t * 1e3:7.2f} us {8 * frames / t / 1e9:5.2f} TB/s")
print(os.environ.get("PGX_GRID_CAP", "8"), " | ".join(res))
