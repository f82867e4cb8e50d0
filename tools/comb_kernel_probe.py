#!/usr/bin/env python3
"""pgx_comb alone (GPU box): HIP-event time per call for a mono 440 Hz comb at one suite block (44 100 frames: one
launch, the reference's loop) and at a look-ahead window of 64 blocks (2 822 400 frames: reduce + apply over time
segments), and for the 512-chain bank; with `pmc` only the launches (for rocprofv3 --pmc passes)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pygmu2_amd as pg
from pygmu2_amd import device
lib = device.ensure_init()
only_launch = "pmc" in sys.argv[1:]
out = {}
for name, batch, n, sr, freqs in (("single_44100", 1, 44100, 44100.0, [440.0]),
                                  ("window_2822400", 1, 2_822_400, 44100.0, [440.0]),
                                  ("bank_512x48000", 512, 48000, 48000.0, [55.0 * 2 ** (i / 96.0) for i in range(512)])):
    rows = int(np.ceil(sr / 20.0)) + 1
    rec = np.zeros(batch, dtype=device.COMB_PARAMS)
    for i, f in enumerate(freqs):
        rec[i] = (0.7, min(max(int(np.rint(sr / max(f, 20.0))), 1), rows - 1), rows)
    params = device.upload_structs(rec)
    x = device.DeviceBuffer.from_host((np.random.default_rng(0).standard_normal((batch, n, 1)) * 0.1).astype(np.float32))
    y = device.DeviceBuffer((batch, n, 1), np.float32)
    ring = device.DeviceBuffer((batch, 2, rows, 1), np.float64, zero=True)
    dmin, dmax = int(rec["delay"].min()), int(rec["delay"].max())
    need = lib.pgx_comb_workspace_bytes(batch, n, 1, dmax, 0)
    ws = device.DeviceBuffer((max(need, 8),), np.uint8)
    st = {"total": 0, "parity": 0}

    def launch():
        device.check(lib.pgx_comb(y.ptr, n, x.ptr, n, batch, n, 1, sr, params.ptr, dmin, dmax, None, None, 20.0, 2400,
                                  ring.ptr, rows, st["total"], st["parity"], None, ws.ptr if need else None))
        st["total"] += n
        st["parity"] ^= 1

    if only_launch:
        for _ in range(5):
            launch()
        device.synchronize()
        continue
    ms = bench.event_avg_ms(launch, 50)
    algo = 8.0 * batch * n
    out[name] = {"us_per_call": round(ms * 1e3, 2), "gsamples_s": round(batch * n / (ms * 1e-3) / 1e9, 2),
                 "algorithmic_gb_s": round(algo / (ms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(algo / (ms * 1e-3) / 1e9 / 8000.0, 4),
                 "segmented": bool(need)}
print(json.dumps(out))
