#!/usr/bin/env python3
"""CombPE throughput (GPU box): the suite protocol of benchmarks/benchmark_pes.py:149-196 (44 100-frame renders,
5 warm-up + 50 timed) on single comb chains, a 512-chain bank under a MixPE (48 000-frame blocks), and the CPU
figure of the same graphs (oracle/seq_kernels.c, gcc -O2, 1 thread).

    python tools/comb_probe.py [nocpu] [nobank]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle.golden_cases import S
import bench_suite as B

sine = lambda f=440.0, a=1.0, ch=1: S("SinePE", frequency=f, amplitude=a, channels=ch)
CONFIGS = [
    ("CombPE (440 Hz, fb 0.7)", S("CombPE", source=sine(), frequency=440.0, feedback=0.7)),
    ("CombPE (220 Hz, fb 0.9, stereo)", S("CombPE", source=sine(330.0, 1.0, 2), frequency=220.0, feedback=0.9)),
    ("CombPE (440 Hz, modulated feedback)", S("CombPE", source=sine(), frequency=440.0,
                                               feedback=sine(2.0, 0.8))),
    ("CombPE (modulated frequency)", S("CombPE", source=sine(),
                                       frequency=S("MixPE", inputs=[S("ConstantPE", value=440.0), sine(0.5, 100.0)]),
                                       feedback=0.7)),
]


def bank_spec(voices):
    return S("MixPE", inputs=[
        S("CombPE", source=S("BlitSawPE", frequency=27.5 * 2 ** (i / 48.0)), frequency=55.0 * 2 ** (i / 96.0),
          feedback=0.7) for i in range(voices)])


def bank_rate(voices=512, frames=48000, steps=20, warm=3):
    import pygmu2_amd as pg
    from pygmu2_amd import device
    import spec_build
    pg.set_sample_rate(48000)
    pe = spec_build.build(bank_spec(voices))
    r = pg.NullRenderer(sample_rate=48000)
    r.set_source(pe)
    r.start()
    for i in range(warm):
        keep = pe.render(i * frames, frames)
    device.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        keep = pe.render((warm + i) * frames, frames)
    keep.dev
    device.synchronize()
    dt = (time.perf_counter() - t0) / steps
    r.stop()
    return dt


def bank_cpu(voices=512, frames=48000, picks=8):
    from oracle import graph_eval
    t_all = 0.0
    idx = [int(i * voices / picks) for i in range(picks)]
    for i in idx:
        g = graph_eval.Node(bank_spec(voices)["inputs"][i], 48000)
        g.render(0, 4800)
        t0 = time.perf_counter()
        g.render(4800, frames)
        t_all += time.perf_counter() - t0
    return t_all / picks * voices


def main():
    with_cpu = "nocpu" not in sys.argv[1:]
    out = {"rows": {}}
    for name, spec in CONFIGS:
        B.SR = 44100
        rates = B.device_rates(spec)
        row = {k: round(v, 1) for k, v in rates.items()}
        if with_cpu:
            row["cpu"] = round(B.cpu_rate(spec, budget_s=1.5), 2)
            row["pipelined_over_cpu"] = round(rates["pipelined"] / row["cpu"], 1)
            row["sync_over_cpu"] = round(rates["sync"] / row["cpu"], 1)
        out["rows"][name] = row
        print(name, row, file=sys.stderr, flush=True)
    if "nobank" not in sys.argv[1:]:
        dt = bank_rate()
        bank = {"ms_per_block": round(dt * 1e3, 4), "msamples_s": round(48000 / dt / 1e6, 3),
                "chain_msamples_s": round(512 * 48000 / dt / 1e6, 1)}
        if with_cpu:
            cpu = bank_cpu()
            bank["cpu_ms_per_block"] = round(cpu * 1e3, 2)
            bank["over_cpu"] = round(cpu / dt, 1)
        out["bank_512_blitsaw_comb_mix_48000"] = bank
    print(json.dumps(out))


if __name__ == "__main__":
    main()
