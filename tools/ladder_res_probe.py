#!/usr/bin/env python3
"""
LadderPE at and above self-oscillation (k = 4 * res * 1.8 passes the small-signal loop gain of 4 at res 0.556,
ladder_pe.py:139-181): the reference's own example (examples/17_ladder_filter.py:43: 800 Hz, res 0.6, drive 1.5)
and stronger settings, under the benchmark_pes.py protocol (44 100-frame renders, 5 + 50), over inputs that do and
do not entrain the oscillating loop.  Prints one JSON line per row: device rates, CPU rate, the PE's segment
statistics (tries, fallbacks, the warm-up it settled on).

    python tools/ladder_res_probe.py [cpu]
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_suite as B
from oracle.golden_cases import S

saw = S("BlitSawPE", frequency=110.0)
sine = S("SinePE", frequency=220.0, amplitude=0.5)
ssaw = S("SuperSawPE", frequency=110.0, voices=7, seed=3)


def ladder(src, f, r, d=1.0, mode="lp24"):
    return S("LadderPE", source=src, frequency=f, resonance=r, mode=mode, drive=d, oversample=2)


ROWS = [
    ("saw110 800Hz res0.3 drive1.5", ladder(saw, 800.0, 0.3, 1.5)),
    ("saw110 800Hz res0.6 drive1.5", ladder(saw, 800.0, 0.6, 1.5)),
    ("saw110 800Hz res0.9 drive1.5", ladder(saw, 800.0, 0.9, 1.5)),
    ("saw110 800Hz res1.0 drive1.5", ladder(saw, 800.0, 1.0, 1.5)),
    ("supersaw110 800Hz res0.6 drive1.5", ladder(ssaw, 800.0, 0.6, 1.5)),
    ("supersaw110 5000Hz res0.9 drive1.0", ladder(ssaw, 5000.0, 0.9, 1.0)),
    ("sine220x0.5 800Hz res0.6 drive1.5", ladder(sine, 800.0, 0.6, 1.5)),
    ("sine220x0.5 800Hz res0.9 drive1.5 (free-running oscillation)", ladder(sine, 800.0, 0.9, 1.5)),
    ("saw110 100Hz res0.5 drive1.0 (slow decay)", ladder(saw, 100.0, 0.5, 1.0)),
]


def main():
    with_cpu = "cpu" in sys.argv[1:]
    only = [a for a in sys.argv[1:] if a != "cpu"]
    from pygmu2_amd import ladder_pe
    for name, spec in ROWS:
        if only and not any(o in name for o in only):
            continue
        stats0 = dict(getattr(ladder_pe, "STATS", {}))
        row = {"row": name}
        row.update({k: round(v, 1) for k, v in B.device_rates(spec).items()})
        stats1 = dict(getattr(ladder_pe, "STATS", {}))
        row["stats"] = {k: stats1[k] - stats0.get(k, 0) for k in stats1}
        if with_cpu:
            row["cpu"] = round(B.cpu_rate(spec, budget_s=1.0), 2)
            row["pipelined_over_cpu"] = round(row["pipelined"] / row["cpu"], 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
