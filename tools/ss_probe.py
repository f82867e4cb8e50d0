#!/usr/bin/env python3
"""SuperSaw mix (512 x 7 oscillators) and C5 voice mix, ms per 48 000-frame block (GPU box)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pygmu2_amd as pg
from pygmu2_amd.sharding import bench_voice_mix
out = {}
only = sys.argv[1:]
for cfg, voices in (("supersaw", 512), ("c5", 512), ("c4", 64)):
    if only and cfg not in only:
        continue
    dt, frames, _, _ = bench_voice_mix(pg, bench._Solo(), 6, 2, voices=voices, config=cfg)
    out[cfg] = round(dt / 6 * 1e3, 4)
print(json.dumps(out))
