#!/usr/bin/env python3
"""C4 (64 x Ladder(SuperSaw 7) -> Mix, 48 000-frame blocks): ms per block over a long stream, with / without the ladder
bank's windows (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pygmu2_amd as pg
from pygmu2_amd import voice_bank
from pygmu2_amd.sharding import bench_voice_mix
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
for flag in (True, False, True):
    voice_bank.LADDER_WINDOWS = flag
    dt, frames, _, _ = bench_voice_mix(pg, bench._Solo(), steps, 5, voices=64, config="c4")
    print(f"windows = {flag}: {dt / steps * 1e3:.4f} ms per block over {steps} blocks", flush=True)
