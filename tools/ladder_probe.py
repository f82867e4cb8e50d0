#!/usr/bin/env python3
"""C4's ladder bank on the GPU box: time per 48 000-frame block and chains that fell back to the sequential
re-render, for several (settle, accurate) warm-up splits."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pygmu2_amd as pg
from pygmu2_amd import device
from pygmu2_amd.sharding import c4_voice

pg.set_sample_rate(48000)
for accurate in (None, 0, 384, 512, 768):
    root = pg.MixPE(*[c4_voice(pg, i) for i in range(64)])
    r = pg.NullRenderer(48000); r.set_source(root); r.start()
    root.render(0, 48000)
    node = root._bank.root
    if accurate is not None:
        node.accurate = accurate
    node.ws.zero_()
    device.synchronize()
    t0 = time.perf_counter()
    for i in range(1, 6):
        keep = root.render(i * 48000, 48000)
    device.synchronize()
    dt = (time.perf_counter() - t0) / 5
    need = node.ws.nbytes
    raw = node.ws.to_host()
    L = device.ensure_init()
    nbytes = L.pgx_ladder_workspace_bytes(64, 48000, 1, node.settle)
    fb = int(raw[nbytes - 16:nbytes - 12].view(np.int32)[0])
    print(json.dumps({"accurate": node.accurate, "settle": node.settle, "ms_per_block": round(dt * 1e3, 4), "fallbacks_in_5_blocks": fb}))
    r.stop()
