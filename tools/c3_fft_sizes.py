#!/usr/bin/env python3
"""pgx_convolve_fft on the C3 filter (65 536 taps, stereo) by transform size: 2^17 (hop 65 537) against 2^18 (hop
196 609) at several render lengths (GPU box, HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pygmu2_amd as pg
for frames in (196_609, 393_218, 786_436, 1_440_000, 2_880_000):
    row = []
    for nfft in (1 << 17, 1 << 18):
        r = bench.conv_fft_roofline(pg, frames, 10, nfft=nfft)
        row.append(f"N=2^{nfft.bit_length() - 1}: {r['avg_launch_ms'] * 1e3:8.2f} us")
    print(f"{frames:9d} frames  " + "   ".join(row), flush=True)
