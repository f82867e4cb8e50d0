#!/usr/bin/env python3
"""SuperSaw mix of 512 instances with / without the bank one block ahead and the mix on the side stream (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pygmu2_amd as pg
from pygmu2_amd import voice_bank
from pygmu2_amd.sharding import bench_voice_mix
for flag in (False, True, False, True):
    voice_bank.PIPELINE_FULL_SUPERSAW_BANK = flag
    dt, frames, _, _ = bench_voice_mix(pg, bench._Solo(), 10, 3, voices=512, config="supersaw")
    print(f"bank ahead + mix on the side stream = {flag}: {dt / 10 * 1e3:.4f} ms per block", flush=True)
