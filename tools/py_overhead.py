#!/usr/bin/env python3
"""Where does the host time of a small-block render go?  (GPU box; cProfile over the C1 loop.)"""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import device
pg.set_sample_rate(44100)
pe = pg.GainPE(pg.SinePE(440.0, 1.0, 0.0, channels=2), gain=0.5)
r = pg.NullRenderer(44100); r.set_source(pe); r.start()
def loop(nblk):
    keep = None
    for i in range(nblk):
        keep = pe.render(i * 1024, 1024)
    device.synchronize()
loop(500)
t0 = time.perf_counter(); loop(4000); dt = time.perf_counter() - t0
print(f"{dt / 4000 * 1e6:.2f} us per block (2 PE calls)")
pr = cProfile.Profile(); pr.enable(); loop(4000); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
