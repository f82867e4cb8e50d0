#!/usr/bin/env python3
"""pgx_supersaw_wide (16 frames per thread) against pgx_supersaw_bank (GPU box): the same bank of `batch` 7-voice
instances rendered block after block by both, states carried; largest difference relative to the peak, and the
HIP-event time per launch of either.  PGX_SSW_WGS_PER_CU / PGX_SS_SEGS shape the segmentation."""
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
blocks = [int(a) for a in sys.argv[2:]] or [48000, 48000, 777, 48000, 100000]
pes = [supersaw_voice(pg, i * (512 // batch)) for i in range(batch)]
nv = 7
params = device.upload_structs(np.concatenate([pe._voice_param_records() for pe in pes]))
init = np.concatenate([pe._voice_initial_state() for pe in pes])
amp = device.DeviceBuffer.from_host(np.array([float(pe._amplitude) for pe in pes], dtype=np.float64))
ref_state = device.DeviceBuffer.from_host(init)
st = [device.DeviceBuffer.from_host(init), device.DeviceBuffer(init.shape, np.float64)]
tab_old = device.DeviceBuffer((lib.pgx_supersaw_bank_table_bytes(batch, nv),), np.uint8)
device.check(lib.pgx_supersaw_bank_tables(tab_old.ptr, batch, nv, 48000.0, params.ptr))
tab = device.DeviceBuffer((lib.pgx_supersaw_wide_table_bytes(batch, nv),), np.uint8)
device.check(lib.pgx_supersaw_wide_tables(tab.ptr, batch, nv, 48000.0, params.ptr))
worst = 0.0
for b, n in enumerate(blocks):
    want = device.DeviceBuffer((batch, n, 1), np.float32)
    got = device.DeviceBuffer((batch, n, 1), np.float32)
    device.check(lib.pgx_supersaw_bank(want.ptr, n, batch, nv, n, 1, 48000.0, params.ptr, ref_state.ptr, amp.ptr))
    device.check(lib.pgx_supersaw_wide(got.ptr, n, batch, nv, n, 1, st[0].ptr, st[1].ptr, amp.ptr, tab.ptr))
    st.reverse()
    w, g = want.to_host().astype(np.float64), got.to_host().astype(np.float64)
    err, peak = float(np.max(np.abs(w - g))), float(np.max(np.abs(w)))
    sdiff = float(np.max(np.abs(ref_state.to_host() - st[0].to_host())))
    worst = max(worst, err / peak)
    print(f"block {b} n {n}: max|d| {err:.3e} of peak {peak:.3f} ({err / peak:.2e}), state difference {sdiff:.2e}, "
          f"segments {lib.pgx_supersaw_wide_segments(batch, nv, n)}", flush=True)
print(f"worst relative difference {worst:.2e}")
for n in (48000, 49152, 24576, 98304):
    out = device.DeviceBuffer((batch, n, 1), np.float32)
    def old():
        device.check(lib.pgx_supersaw_bank_seg(out.ptr, n, batch, nv, n, 1, 48000.0, params.ptr, st[0].ptr, st[1].ptr,
                                               amp.ptr, tab_old.ptr))
    def new():
        device.check(lib.pgx_supersaw_wide(out.ptr, n, batch, nv, n, 1, st[0].ptr, st[1].ptr, amp.ptr, tab.ptr))
    t_old, t_new = bench.event_avg_ms(old, 20) * 1e3, bench.event_avg_ms(new, 20) * 1e3
    print(f"batch {batch} n {n:6d}: bank_seg {t_old:7.2f} us ({lib.pgx_supersaw_bank_segments(batch, n)} segments), "
          f"wide {t_new:7.2f} us ({lib.pgx_supersaw_wide_segments(batch, nv, n)} segments)", flush=True)
