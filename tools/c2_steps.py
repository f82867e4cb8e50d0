#!/usr/bin/env python3
"""Where a short C2 stream spends its time (GPU box): host time of every timed step of `bench.py --steps 20
--warmup 5` (the driver's run), then the device wait."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pygmu2_amd as pg
from pygmu2_amd import device
steps, warmup, frames = 20, 5, 1_000_000
for rep in range(2):
    pe, r = bench.c2_graph(pg)
    origin = (warmup + 1000) * frames
    for i in range(warmup):
        pos = i * frames if i < warmup - 1 else origin - frames
        keep = pe.render(pos, frames)
    device.synchronize()
    t = [time.perf_counter()]
    for i in range(steps):
        keep = pe.render(origin + i * frames, frames)
        t.append(time.perf_counter())
    device.synchronize()
    t.append(time.perf_counter())
    r.stop()
    us = [(b - a) * 1e6 for a, b in zip(t, t[1:])]
    print(f"rep {rep}: total {sum(us):.1f} us = {sum(us) / steps:.2f} us/step; host per step:",
          " ".join(f"{u:.1f}" for u in us[:-1]), f"| final wait {us[-1]:.1f}")
