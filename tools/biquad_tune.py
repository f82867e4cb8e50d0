#!/usr/bin/env python3
"""HIP-event timing of pgx_biquad_const (settled vs exact) at a few sizes; compares outputs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import device
from pygmu2_amd.biquad_pe import rbj_coefficients, settle_frames

lib = device.ensure_init()
pg.set_sample_rate(44100)
c = rbj_coefficients(pg.BiquadMode.LOWPASS, 1000.0, 0.707, 0.0, 44100.0)
coef = device.DeviceBuffer.from_host(np.asarray(c, dtype=np.float64))
W = settle_frames(c[3], c[4])
tag = f"WGS={os.environ.get('PGX_BQ_WGS', '-')} SPW={os.environ.get('PGX_BQ_SEG_PER_WARM', '-')}"
for frames, reps in ((1_000_000, 200), (1 << 22, 100), (1 << 24, 30), (1 << 26, 10)):
    x = pg.SinePE(440.0).render(0, frames).dev
    outs = {}
    tables = device.DeviceBuffer((lib.pgx_biquad_table_doubles(),), np.float64)
    device.check(lib.pgx_biquad_tables(tables.ptr, coef.ptr, 1))
    for name, settle in (("settled", W), ("exact", 0)):
        out = device.DeviceBuffer((frames, 1), np.float32)
        state = device.DeviceBuffer((1, 2), np.float64, zero=True)
        ws = device.DeviceBuffer((max(lib.pgx_biquad_workspace_bytes(1, frames, 1, settle), 1),), np.uint8)
        def launch():
            device.check(lib.pgx_biquad_const(out.ptr, 0, x.ptr, 0, 1, frames, 1, coef.ptr, tables.ptr if settle else None, settle, state.ptr, ws.ptr))
        for _ in range(3):
            launch()
        e0, e1 = device.Event(), device.Event()
        e0.record()
        for _ in range(reps):
            launch()
        e1.record()
        ms = e1.elapsed_ms_since(e0) / reps
        state.zero_()
        launch()
        outs[name] = out.to_host()
        print(f"{tag} frames={frames:9d} {name:8s} {ms*1e3:9.2f} us  {8.0*frames/(ms*1e-3)/1e9:8.1f} GB/s algorithmic")
    d = np.abs(outs["settled"].astype(np.float64) - outs["exact"])
    print(f"   settled vs exact: max|d|={d.max():.3e}  differing={int((outs['settled'] != outs['exact']).sum())}")
