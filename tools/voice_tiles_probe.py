#!/usr/bin/env python3
"""
The on-chip mix (pgx_voice_tiles) against the layered path it replaces, on the C5 graph: agreement and time per block.

    python tools/voice_tiles_probe.py [voices=512] [blocks=24] [frames=48000]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pygmu2_amd as pg
from pygmu2_amd import voice_bank, device
from pygmu2_amd.sharding import c5_voice

voices = int(sys.argv[1]) if len(sys.argv) > 1 else 512
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 24
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 48000
pg.set_sample_rate(48000)


def run(tiles: bool, keep: int):
    voice_bank.VOICE_TILES = tiles
    mix = pg.MixPE(*[c5_voice(pg, i * (512 // voices)) for i in range(voices)])
    got = []
    with pg.NullRenderer(sample_rate=48000) as r:
        r.set_source(mix)
        r.start()
        for b in range(3):                                   # warm-up: tables, pools, the walk one block ahead
            s = mix.render(b * frames, frames)
            if b < keep:
                got.append(s.data.copy())
        device.synchronize()
        t0 = time.perf_counter()
        for b in range(3, 3 + blocks):
            s = mix.render(b * frames, frames)
        host = (time.perf_counter() - t0) / blocks
        device.synchronize()
        dt = (time.perf_counter() - t0) / blocks
        print(f"   host enqueue {host * 1e6:.1f} us per block", end="")
        got.append(s.data.copy())
    return dt, got


for tiles in (False, True, False, True):
    dt, got = run(tiles, 3)
    print(f"tiles={tiles}: {dt * 1e6:8.1f} us per block", flush=True)
    if tiles:
        new = got
    else:
        old = got
peak = max(float(np.max(np.abs(a))) for a in old)
for i, (a, b) in enumerate(zip(old, new)):
    print(f"block {i if i < 3 else 2 + blocks}: max |new - old| / peak = {float(np.max(np.abs(a - b))) / peak:.3e}")
