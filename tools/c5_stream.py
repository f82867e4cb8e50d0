#!/usr/bin/env python3
"""A C5 stream as shipped (512 voices, 48 000-frame blocks): argv[1] blocks (default 63 = 1 + 2 + 4 + 8 x 7: whole windows).
For `tools/pmc_pass.sh <name> FETCH_SIZE|WRITE_SIZE tools/c5_stream.py`: the run's total bytes over its blocks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import device
from pygmu2_amd.sharding import c5_voice
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 63
pg.set_sample_rate(48000)
mix = pg.MixPE(*[c5_voice(pg, i) for i in range(512)])
with pg.NullRenderer(sample_rate=48000) as r:
    r.set_source(mix)
    r.start()
    for b in range(blocks):
        keep = mix.render(b * 48000, 48000)
    device.synchronize()
print("blocks", blocks)
