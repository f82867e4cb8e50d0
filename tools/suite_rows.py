#!/usr/bin/env python3
"""Device rates of chosen benchmark_pes.py rows (substring match on the config name), GPU box."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_suite as B
want = sys.argv[1:]
for name, spec in B.CONFIGS:
    if want and not any(w in name for w in want):
        continue
    print(f"{name:40s}", json.dumps({k: round(v, 1) for k, v in B.device_rates(spec).items()}), flush=True)
