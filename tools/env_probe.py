#!/usr/bin/env python3
"""EnvelopePE attack/release follower (k_env_newton) at benchmark_pes.py's 44 100-frame renders and at the
autowah script's 1024-frame blocks; run under tools/kernel_trace.sh for per-kernel durations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pygmu2_amd as pg
from pygmu2_amd import device

SR = 44100
pg.set_sample_rate(SR)
rng = np.random.default_rng(0)
sources = {
    "sine 440": lambda: pg.SinePE(frequency=440.0),
    "noise": lambda: pg.ArrayPE((rng.standard_normal((SR * 2, 1)) * 0.3).astype(np.float32)),
}
for name, make in sources.items():
    for block, reps in ((44100, 40), (1024, 80)):
        env = pg.EnvelopePE(make(), attack=0.01, release=0.1)
        r = pg.NullRenderer(sample_rate=SR)
        r.set_source(env)
        r.start()
        for i in range(3):
            keep = env.render(0, block)
        device.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            keep = env.render(0, block)
        device.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{name:10s} block {block:6d}: {dt * 1e6:8.1f} us / render  {block / dt * 1e-6:8.1f} Msamples/s")
        r.stop()
