#!/usr/bin/env python3
"""The stand-alone AdsrGatedPE rows of the bench line (44 100-frame renders, 5 + 50), GPU box."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_suite as B
from oracle.golden_cases import S
for hz in (2.0, 7.0):
    spec = S("AdsrGatedPE", gate=S("PeriodicGate", frequency=hz, duty_cycle=0.5), attack_time=0.01, decay_time=0.1,
             sustain_level=0.7, release_time=0.2)
    row = {k: round(v, 1) for k, v in B.device_rates(spec).items()}
    row["cpu"] = round(B.cpu_rate(spec, budget_s=1.0), 2)
    print(hz, json.dumps(row), flush=True)
# a long stream: 400 renders, so that the look-ahead windows reach their largest size
import time
import pygmu2_amd as pg, spec_build
from pygmu2_amd import device
pg.set_sample_rate(44100)
pe = spec_build.build(spec)
r = pg.NullRenderer(sample_rate=44100); r.set_source(pe); r.start()
for i in range(40):
    pe.render(i * 44100, 44100)
device.synchronize()
t0 = time.perf_counter()
for i in range(40, 1240):
    keep = pe.render(i * 44100, 44100)
keep.dev
device.synchronize()
dt = time.perf_counter() - t0
print(f"1200 renders of 44 100 frames, pipelined: {1200 * 44100 / dt / 1e6:.0f} Msamples/s")
