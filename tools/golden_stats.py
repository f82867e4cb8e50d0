#!/usr/bin/env python3
"""HIP path vs the reference-generated golden blocks: how many float32 samples are bit-identical, and the worst
error relative to a block's peak (the figures quoted in DESIGN.md section 6)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from spec_build import run_case

cases = json.load(open(os.path.join(ROOT, "tests", "golden", "cases.json")))
cases = cases["cases"] if isinstance(cases, dict) else cases
golden = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
total = same = 0
worst = (0.0, "")
quiet = (0.0, "")
for case in cases:
    outs = run_case(case)
    for i in case["keep"]:
        want = golden[f"{case['name']}/{i}"]
        got = outs[i]
        total += want.size
        same += int(np.sum(got.view(np.uint32) == want.view(np.uint32)))
        peak = float(np.max(np.abs(want))) if want.size else 0.0
        if not np.all(np.isfinite(want)):
            continue
        err = float(np.max(np.abs(got.astype(np.float64) - want))) if want.size else 0.0
        if peak > 1e-3:                      # (near-silent blocks are held to the 1e-7 absolute floor instead)
            if err / peak > worst[0]:
                worst = (err / peak, case["name"])
        elif err > quiet[0]:
            quiet = (err, case["name"])
print(f"{len(cases)} cases, {total} samples, {same} bit-identical ({100.0 * same / total:.1f} %), "
      f"worst max|d|/peak = {worst[0]:.2e} ({worst[1]}); worst max|d| on a near-silent block = {quiet[0]:.2e} ({quiet[1]})")
