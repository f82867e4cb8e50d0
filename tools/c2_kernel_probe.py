#!/usr/bin/env python3
"""pgx_biquad_sine alone: HIP-event time per launch at 1 M / 33 M / 2^26 frames (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pygmu2_amd as pg
for frames, launches in ((1_000_000, 200), (33_000_000, 50), (1 << 26, 20)):
    r = bench.biquad_sine_roofline(pg, frames, launches, 10 ** 9)
    print(frames, round(r["avg_launch_ms"] * 1e3, 2), "us", r["achieved"], "GB/s", r["gsamples_per_s"], "Gsamples/s")
