#!/usr/bin/env python3
"""Print, per golden case, how close the HIP path is to the reference golden vectors:
samples that differ bit-wise, worst error relative to the block peak.  (GPU box only;
diagnostic, not a test.)"""

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from spec_build import run_case  # noqa: E402

cases = json.load(open(os.path.join(HERE, "golden", "cases.json")))
G = np.load(os.path.join(HERE, "golden", "golden.npz"))
tot = diff = 0
for c in cases:
    outs = run_case(c)
    worst, nd, ns = 0.0, 0, 0
    for i in c["keep"]:
        g = G[f"{c['name']}/{i}"]
        o = outs[i]
        ns += g.size
        nd += int(np.sum(o != g))
        pk = max(float(np.max(np.abs(g))), 1e-30)
        worst = max(worst, float(np.max(np.abs(o.astype(np.float64) - g))) / pk)
    tot += ns
    diff += nd
    print(f"{c['name']:34s} samples={ns:8d} differ={nd:7d} worst_rel_to_peak={worst:.2e}")
print(f"TOTAL samples={tot} differ={diff} ({100.0 * diff / tot:.4f} %)")
