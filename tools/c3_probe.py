#!/usr/bin/env python3
"""C3 on the GPU box: pgx_convolve_fft alone (HIP events) at the three bench shapes, and the ConvolvePE step."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pygmu2_amd as pg
out = {}
for frames, launches in ((96_000, 50), (65_537, 50), (1_440_000, 10)):
    r = bench.conv_fft_roofline(pg, frames, launches)
    out[str(frames)] = {"avg_launch_us": round(r["avg_launch_ms"] * 1e3, 2), "GB/s": r["achieved"]}
dt, f = bench.bench_c3(pg, bench._Solo(), 20, 3)
out["c3_step_us"] = round(dt / 20 * 1e6, 2)
out["c3_msamples_s"] = round(f * 20 / dt / 1e6, 1)
print(json.dumps(out))
