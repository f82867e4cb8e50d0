#!/usr/bin/env python3
"""What a kernel that ONLY writes reaches on this part (GPU box): pgx_memset (hipMemsetAsync) and the library's own fill
kernel over buffers of the C2 window's size, HIP events over 200 launches after 50 -- the ceiling the fused C2 kernel
(4 B/frame written, nothing read) is priced against besides the data-sheet 8 TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import pygmu2_amd as pg
from pygmu2_amd import device
lib = device.ensure_init()
pg.set_sample_rate(44100)
for frames in (33_000_000, 134_000_000):
    buf = device.DeviceBuffer((frames, 1), np.float32)
    ms = bench.event_avg_ms(lambda: device.check(lib.pgx_memset(buf.ptr, 0, buf.nbytes), "pgx_memset"), 200, 50)
    print(f"pgx_memset {buf.nbytes / 1e6:7.1f} MB: {ms * 1e3:7.2f} us = {buf.nbytes / (ms * 1e-3) / 1e12:.2f} TB/s")
    src = pg.ConstantPE(0.25)
    r = pg.NullRenderer(sample_rate=44100); r.set_source(src); r.start()
    ms = bench.event_avg_ms(lambda: src._render(0, frames), 100, 20)
    r.stop()
    print(f"ConstantPE render {frames * 4 / 1e6:7.1f} MB: {ms * 1e3:7.2f} us = {frames * 4 / (ms * 1e-3) / 1e12:.2f} TB/s (fill kernel + allocation)")
