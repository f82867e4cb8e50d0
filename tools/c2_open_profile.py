#!/usr/bin/env python3
"""What opening a look-ahead window of the C2 chain costs on the host (GPU box): cProfile over 300 window openings."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pygmu2_amd as pg
from pygmu2_amd import device, look_ahead
pe, r = bench.c2_graph(pg)
frames = 1_000_000
pos = 0
def stream(k):
    global pos
    for _ in range(k):
        pe.render(pos, frames)
        pos += frames
stream(3)
device.synchronize()
look_ahead.AHEAD_BLOCKS = 2          # every second pull opens a window: the opening dominates
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
stream(600)
pr.disable()
device.synchronize()
dt = time.perf_counter() - t0
print(f"{dt / 300 * 1e6:.1f} us per window of 2 blocks (host + device)")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
