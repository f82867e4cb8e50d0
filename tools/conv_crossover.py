#!/usr/bin/env python3
"""Time ConvolvePE's two device paths (direct MFMA vs FFT) over filter lengths; stereo, 96 000-frame blocks."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import convolve_pe, device

pg.set_sample_rate(48000)
rng = np.random.default_rng(0)
n = 96_000
x = pg.ArrayPE((rng.standard_normal((n, 2)) * 0.1).astype(np.float32))
for L in (512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072):
    h = pg.ArrayPE((rng.standard_normal(L) * np.exp(-np.arange(L) / (L / 8))).astype(np.float32))
    res = {}
    for name, thr in (("direct", 1 << 30), ("fft", 1)):
        convolve_pe.FFT_MIN_TAPS = thr
        pe = pg.ConvolvePE(x, h)
        for _ in range(3):
            pe.render(0, n)
        device.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            keep = pe.render(0, n)
        device.synchronize()
        res[name] = (time.perf_counter() - t0) / reps * 1e6
    print(f"L={L:7d}  direct {res['direct']:9.1f} us   fft {res['fft']:9.1f} us", flush=True)
