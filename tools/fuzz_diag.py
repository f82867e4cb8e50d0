#!/usr/bin/env python3
"""Re-run seeds of tests/test_gpu_fuzz.py and show, per block, where the HIP path and the oracle part."""
import sys, json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from oracle.graph_eval import run_case as oracle_run
from spec_build import run_case as hip_run
def kinds(g):
    out=[g["pe"]]
    for k,v in g.items():
        if isinstance(v, dict) and "pe" in v: out += kinds(v)
        if k=="inputs":
            for x in v: out += kinds(x)
    return out
for seed in [int(a) for a in sys.argv[1:]]:
    c = m._graph(seed)
    got = hip_run(c); want = oracle_run(c)
    print(seed, c["sr"], json.dumps(c["graph"])[:900])
    for i,(g,w) in enumerate(zip(got,want)):
        d=np.abs(g.astype(np.float64)-w); err=float(d.max()); peak=float(np.max(np.abs(w)))
        print("   block", c["blocks"][i], "err %.3e peak %.3e first bad idx %s" % (err, peak, np.argmax(d > 1e-5*peak+1e-6) if err>1e-5*peak+1e-6 else "-"))
