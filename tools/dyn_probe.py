#!/usr/bin/env python3
"""CompressorPE / LimiterPE / ExpanderPE / WindowPE at benchmark_pes.py's 44 100-frame renders; run under
tools/kernel_trace.sh for the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import device

SR = 44100
pg.set_sample_rate(SR)
for name, make in (("CompressorPE", lambda: pg.CompressorPE(pg.SinePE(frequency=440.0))),
                   ("LimiterPE", lambda: pg.LimiterPE(pg.SinePE(frequency=440.0))),
                   ("WindowPE", lambda: pg.WindowPE(pg.SinePE(frequency=440.0)))):
    pe = make()
    r = pg.NullRenderer(sample_rate=SR)
    r.set_source(pe)
    r.start()
    for i in range(3):
        keep = pe.render(i * SR, SR)
    device.synchronize()
    t0 = time.perf_counter()
    for i in range(3, 33):
        keep = pe.render(i * SR, SR)
    device.synchronize()
    print(f"{name:14s} {(time.perf_counter() - t0) / 30 * 1e6:8.1f} us / render")
    r.stop()
