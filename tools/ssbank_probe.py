#!/usr/bin/env python3
"""pgx_supersaw_bank_seg alone (GPU box): HIP-event time per launch for a bank of `batch` 7-voice instances, by block
length and with / without the per-voice tables.  PGX_SS_SEGS=k forces the segment count."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pygmu2_amd as pg
from pygmu2_amd import device
from pygmu2_amd.sharding import supersaw_voice
lib = device.ensure_init()
pg.set_sample_rate(48000)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pes = [supersaw_voice(pg, i * (512 // batch)) for i in range(batch)]
nv = 7
params = device.upload_structs(np.concatenate([pe._voice_param_records() for pe in pes]))
init = np.concatenate([pe._voice_initial_state() for pe in pes])
st_a, st_b = device.DeviceBuffer.from_host(init), device.DeviceBuffer(init.shape, np.float64)
amp = device.DeviceBuffer.from_host(np.array([float(pe._amplitude) for pe in pes], dtype=np.float64))
tables = device.DeviceBuffer((lib.pgx_supersaw_bank_table_bytes(batch, nv),), np.uint8)
device.check(lib.pgx_supersaw_bank_tables(tables.ptr, batch, nv, 48000.0, params.ptr))
for n in (48000, 49152, 45056, 24576, 12288, 98304):
    out = device.DeviceBuffer((batch, n, 1), np.float32)
    row = []
    for tab in (tables.ptr, None):
        def launch():
            device.check(lib.pgx_supersaw_bank_seg(out.ptr, n, batch, nv, n, 1, 48000.0, params.ptr, st_a.ptr, st_b.ptr,
                                                   amp.ptr, tab))
        row.append(bench.event_avg_ms(launch, 20) * 1e3)
    print(f"batch {batch} n {n:6d} segs {lib.pgx_supersaw_bank_segments(batch, n)}: {row[0]:7.2f} us with tables, "
          f"{row[1]:7.2f} us without", flush=True)
