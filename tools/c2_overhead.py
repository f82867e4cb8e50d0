#!/usr/bin/env python3
"""Where does the host time of a C2 step go?  (GPU box; cProfile over 1 M-frame steps through look-ahead windows.)"""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import device
pg.set_sample_rate(44100)
pe = pg.BiquadPE(pg.SinePE(440.0), 1000.0, 0.707)
r = pg.NullRenderer(44100); r.set_source(pe); r.start()
N = 1_000_000
def loop(nblk, base):
    keep = None
    for i in range(nblk):
        keep = pe.render((base + i) * N, N)
    device.synchronize()
    return base + nblk
pos = loop(64, 0)
t0 = time.perf_counter(); pos = loop(320, pos); dt = time.perf_counter() - t0
print(f"{dt / 320 * 1e6:.2f} us per 1M-frame step = {N * 320 / dt / 1e6:.0f} Msamples/s")
pr = cProfile.Profile(); pr.enable(); pos = loop(320, pos); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
