#!/usr/bin/env python3
"""A rank's share of C5 (64 voices at eight ranks): host time to ENQUEUE a block against the time a block takes, and
where the host time goes (cProfile).  GPU box."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import device
from pygmu2_amd.sharding import c5_voice, shard_indices
pg.set_sample_rate(48000)
block = 48000
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
root = pg.MixPE(*[c5_voice(pg, i) for i in shard_indices(512, 0, world)])
r = pg.NullRenderer(sample_rate=48000); r.set_source(root); r.start()
pos = 0
for i in range(20):
    root.render(pos, block); pos += block
device.synchronize()
host = 0.0
t0 = time.perf_counter()
for i in range(200):
    a = time.perf_counter()
    keep = root.render(pos, block); pos += block
    host += time.perf_counter() - a
    if i % 10 == 9:                      # keep the queue short: the host must not be measured waiting for a full queue
        device.synchronize()
device.synchronize()
total = time.perf_counter() - t0
print(f"world={world}: host enqueue {host / 200 * 1e6:.1f} us per block (with a device wait every 10 blocks: "
      f"{total / 200 * 1e6:.1f} us per block in all)")
t0 = time.perf_counter()
for i in range(200):
    keep = root.render(pos, block); pos += block
device.synchronize()
print(f"pipelined: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per block")
pr = cProfile.Profile(); pr.enable()
for i in range(200):
    keep = root.render(pos, block); pos += block
pr.disable()
device.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
