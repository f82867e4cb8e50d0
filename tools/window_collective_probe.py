"""one rank's share with the 1-rank communicator in the loop: per-block collectives against one per window"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import comm, device, sharding
from pygmu2_amd.sharding import ShardedMixPE, RcclReducer, mix_voice_factory
device.ensure_init()
comm.init(0, 1, comm.unique_id())
pg.set_sample_rate(48000)
block = 48000
config = sys.argv[1] if len(sys.argv) > 1 else "supersaw"      # supersaw | c4 | c5
make, voices = mix_voice_factory(config)
for world in (8, 4, 2):
    for whole in (True, False):
        sharding.WINDOW_COLLECTIVES = whole
        root = ShardedMixPE([make(pg, i) for i in range(voices)], 0, world, reducer=RcclReducer())
        r = pg.NullRenderer(48000); r.set_source(root); r.start()
        keep = None
        for i in range(15):             # (the windows of 2, 4, 8 blocks open, their buffers are allocated)
            keep = root.render(i * block, block)
        keep.dev
        device.synchronize()
        c0 = root._reducer.calls
        t0 = time.perf_counter()
        reps = 48
        for i in range(reps):
            keep = root.render((15 + i) * block, block)
        keep.dev
        device.synchronize()
        dt = (time.perf_counter() - t0) / reps
        r.stop()
        print(f"{config} world={world} window collectives={whole}: {dt*1e6:7.1f} us per block, {root._reducer.calls - c0} collectives for {reps} blocks", flush=True)
