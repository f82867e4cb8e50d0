#!/usr/bin/env python3
"""C4 (64 x Ladder(SuperSaw 7) -> Mix, 48 000-frame blocks) over a long stream: ms per block by the largest window of
the ladder bank (voice_bank.LADDER_WINDOW_MAX blocks) and the lane count of k_ladder_segments (PGX_LADDER_LANES is read
once per process: one process per lane count).  usage: c4_windows_probe.py [max_blocks ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pygmu2_amd as pg
from pygmu2_amd import voice_bank
from pygmu2_amd.sharding import bench_voice_mix
sizes = [int(a) for a in sys.argv[1:]] or [8, 16, 32]
voice_bank.LADDER_WINDOW_FRAMES = 1 << 22
for config in ("c4", "c4r06"):
    for m in sizes:
        voice_bank.LADDER_WINDOW_MAX = m
        warm = 2 * m - 1                      # 1 + 2 + 4 + ... + m / 2 ... the windows up to m open in the warm-up
        steps = 3 * m if m <= 16 else 2 * m   # whole windows of m
        dt, frames, _, _ = bench_voice_mix(pg, bench._Solo(), steps, warm, voices=64, config=config)
        print(f"{config}: windows up to {m:2d} blocks, lanes {os.environ.get('PGX_LADDER_LANES', 'model')}: "
              f"{dt / steps * 1e3:.4f} ms per block over {steps} blocks", flush=True)
