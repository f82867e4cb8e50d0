#!/usr/bin/env python3
"""What one rank of a G-way sharded C5 run (argv: `supersaw` = the 512-voice SuperSaw mix) does -- its
512/G voices, no collective: block time on one GPU.
The per-voice chains are sequential in time, so this is the floor the RCCL version can reach."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import device
from pygmu2_amd.sharding import c4_voice, c5_voice, shard_indices, supersaw_voice

pg.set_sample_rate(48000)
block = 48000
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
make, total = {"supersaw": (supersaw_voice, 512), "c5": (c5_voice, 512), "c4": (c4_voice, 64)}[which]
for world in [int(w) for w in os.environ.get("PGX_WORLDS", "1,2,4,8").split(",")]:
    voices = [make(pg, i) for i in shard_indices(total, 0, world)]
    root = pg.MixPE(*voices)
    if which == "c4" and world > 1:
        root.__dict__["_mix_windows"] = True      # as ShardedMixPE asks: windows at the level of the mix (one collective each)
    r = pg.NullRenderer(sample_rate=48000)
    r.set_source(root)
    r.start()
    # (15 blocks of warm-up: a small bank's windows of 2, 4, 8 blocks open -- and their buffers are allocated, a hipMalloc
    # of a few hundred MB is milliseconds on some boxes -- before the clock starts; 24 timed blocks = three whole windows)
    # PGX_SHARD_REPS: timed blocks (default 24 / 64; a few hundred -- 384, 256: whole windows -- let the chip reach the clock
    # a stream runs at: the same kernels are 5 - 10 % faster than in a 24-block burst)
    warm = 63 if which == "c4" else 15            # (C4's windows grow to 32 blocks)
    for i in range(warm):
        root.render(i * block, block)
    device.synchronize()
    t0 = time.perf_counter()
    reps = int(os.environ.get("PGX_SHARD_REPS", "0")) or (64 if which == "c4" else 24)
    for i in range(reps):
        keep = root.render((warm + i) * block, block)
    device.synchronize()
    dt = (time.perf_counter() - t0) / reps
    r.stop()
    print(f"world={world}: {len(voices):3d} voices on this rank, {dt * 1e6:7.1f} us per 48000-frame block "
          f"-> {block / dt / 1e6:6.1f} Msamples/s before the all-reduce", flush=True)
