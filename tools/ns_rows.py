#!/usr/bin/env python3
"""bench.py's north_star_pe_rows alone (GPU box): CombPE / LadderPE / AdsrGatedPE under the suite protocol."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pygmu2_amd as pg
print(json.dumps(bench.north_star_pe_rows(pg, "nocpu" not in sys.argv[1:]), indent=1))
