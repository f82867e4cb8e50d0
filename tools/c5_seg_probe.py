#!/usr/bin/env python3
"""C5 at one GPU (512 voices): ms per block, and the chain kernel alone (HIP events) -- with the chain in time segments
when PGX_BBW_MAX_BATCH=512 PGX_BBW_WGS=<workgroups> say so."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pygmu2_amd as pg
from pygmu2_amd.sharding import bench_voice_mix
r = bench.bank_kernel_fp64(pg, "c5")
dt, frames, _, _ = bench_voice_mix(pg, bench._Solo(), 50, 5, voices=512, config="c5")
print(f"MAX_BATCH {os.environ.get('PGX_BBW_MAX_BATCH')} WGS {os.environ.get('PGX_BBW_WGS')}: C5 {dt / 50 * 1e3:.4f} ms per block; "
      f"chain node alone {r['avg_launch_ms'] * 1e3 if r else None} us")
