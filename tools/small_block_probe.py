#!/usr/bin/env python3
"""Small-block streaming rows (GPU box): C1, the hello-sine example, the two autowah graphs -- Msamples/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pygmu2_amd as pg
from pygmu2_amd import processing_element as P, look_ahead, read_ahead
dt, f = bench.bench_c1(pg, bench._Solo(), 5, 1)
print(f"fast paths {P.FAST_PATHS}, look-ahead up to {look_ahead.AHEAD_BLOCKS} blocks, read-ahead up to {read_ahead.AHEAD_BLOCKS}: "
      f"C1 {f * 5 / dt / 1e6:.0f}, hello sine {bench.hello_sine_case(pg):.0f}, autowah biquad "
      f"{bench.autowah_case(pg, 'biquad'):.0f}, autowah svf {bench.autowah_case(pg, 'svf'):.0f} Msamples/s", flush=True)
