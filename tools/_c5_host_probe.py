import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import device
from pygmu2_amd.sharding import c5_voice, shard_indices
pg.set_sample_rate(48000)
block = 48000
voices = [c5_voice(pg, i) for i in shard_indices(512, 0, 8)]
root = pg.MixPE(*voices)
r = pg.NullRenderer(sample_rate=48000); r.set_source(root); r.start()
for i in range(5):
    root.render(i * block, block)
device.synchronize()
# host time alone: enqueue 100 blocks, time the enqueue, then the drain
t0 = time.perf_counter()
for i in range(100):
    keep = root.render((5 + i) * block, block)
t1 = time.perf_counter()
device.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e4*(t1-t0):.1f} us per block, drain after {1e4*(t2-t1):.1f} us per block", flush=True)
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    keep = root.render((105 + i) * block, block)
pr.disable()
device.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
