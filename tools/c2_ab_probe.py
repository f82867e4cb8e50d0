#!/usr/bin/env python3
"""pgx_biquad_sine by HIP events at the bench's launch sizes (GPU box); run once per library build (PGX_LIB_PATH) to
compare variants: three passes, alternating sizes, the median per size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import pygmu2_amd as pg
res = {}
for rep in range(3):
    for frames, launches in ((134_000_000, 200), (33_000_000, 300), (1 << 26, 200), (16_000_000, 300)):      # (streams: the chip at its held clock)
        r = bench.biquad_sine_roofline(pg, frames, launches, 10 ** 9, warm=50)
        res.setdefault(frames, []).append(r["avg_launch_ms"] * 1e3)
print(os.environ.get("PGX_LIB_PATH", "default build"), {k: round(float(np.median(v)), 2) for k, v in res.items()})
