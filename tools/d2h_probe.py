#!/usr/bin/env python3
"""C2 with the root Snippet read on the host every step (`.data`: stream sync + device-to-host copy into a numpy
array): the PCIe-inclusive rate next to bench.py's HBM-resident `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import device

pg.set_sample_rate(44100)
pe = pg.BiquadPE(pg.SinePE(frequency=440.0), frequency=1000.0, q=0.707, mode=pg.BiquadMode.LOWPASS)
r = pg.NullRenderer(sample_rate=44100)
r.set_source(pe)
r.start()
frames = 1_000_000
for mode in ("resident", "host copy"):
    for i in range(5):
        s = pe.render(i * frames, frames)
        if mode != "resident":
            s.data
    device.synchronize()
    t0 = time.perf_counter()
    steps = 50
    for i in range(5, 5 + steps):
        s = pe.render(i * frames, frames)
        if mode != "resident":
            a = s.data
    device.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{mode:10s}: {dt * 1e6:8.1f} us per 1M-frame step = {frames / dt / 1e6:9.1f} Msamples/s"
          + (f"  ({frames * 4 / dt / 1e9:.1f} GB/s over PCIe)" if mode != "resident" else ""))
