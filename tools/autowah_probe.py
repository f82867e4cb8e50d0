#!/usr/bin/env python3
"""The autowah graphs of benchmarks/profile_biquad_vs_svfilter.py (BASELINE config 1's script), streamed in
1024-frame blocks for 8 s like the script does, and in 44 100-frame blocks like benchmark_pes.py; plus
BiquadPE(SinePE) streamed in 1024-frame blocks.  Prints Msamples/s; `cpu` adds the oracle on this host."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import device, transforms as tf

SR, SECONDS = 44100, 8


def graph(kind):
    src = pg.SinePE(frequency=220.0, amplitude=0.8)
    env = pg.EnvelopePE(src, attack=0.005, release=0.05, mode=pg.DetectionMode.PEAK)
    ctl = pg.TransformPE(env, func=tf.Chain(tf.Clip(0.0, 1.0), tf.Sqrt(), tf.Affine(2900.0, 100.0)), name="env_to_freq")
    flt = (pg.BiquadPE if kind == "biquad" else pg.SVFilterPE)(src, frequency=ctl, q=10.0, mode=pg.BiquadMode.LOWPASS)
    return pg.GainPE(flt, gain=1.0)


def run(make, block, total):
    pg.set_sample_rate(SR)
    root = pg.CropPE(make(), 0, total)
    r = pg.NullRenderer(sample_rate=SR)
    r.set_source(root)
    r.start()
    for warm in range(2):
        root.render(warm * block, block)
    r.stop()
    r.start()
    device.synchronize()
    t0 = time.perf_counter()
    pos = 0
    while pos < total:
        n = min(block, total - pos)
        keep = r.render(pos, n)
        pos += n
    device.synchronize()
    dt = time.perf_counter() - t0
    r.stop()
    return total / dt / 1e6


total = SR * SECONDS
for name, make in (("autowah_biquad", lambda: graph("biquad")), ("autowah_svf", lambda: graph("svf")),
                   ("biquad_on_sine", lambda: pg.BiquadPE(pg.SinePE(440.0), 1000.0, 0.707))):
    for block in (1024, 44100):
        print(f"{name:16s} block={block:6d}  {run(make, block, total):9.2f} Msamples/s", flush=True)

if "cpu" in sys.argv:
    from oracle import graph_eval
    from oracle.golden_cases import S
    def spec(kind):
        src = S("SinePE", frequency=220.0, amplitude=0.8)
        env = S("EnvelopePE", source=src, attack=0.005, release=0.05, mode="peak")
        ctl = S("TransformPE", source=env, ops=[["clip", 0.0, 1.0], ["sqrt"], ["affine", 2900.0, 100.0]])
        return S("GainPE", source=S(kind, source=src, frequency=ctl, q=10.0, mode="lowpass"), gain=1.0)
    for name, sp in (("autowah_biquad", spec("BiquadPE")), ("autowah_svf", spec("SVFilterPE")),
                     ("biquad_on_sine", S("BiquadPE", source=S("SinePE", frequency=440.0), frequency=1000.0, q=0.707))):
        for block in (1024, 44100):
            g = graph_eval.Node(sp, SR)
            t0 = time.perf_counter()
            pos = 0
            while pos < total:
                n = min(block, total - pos)
                g.render(pos, n)
                pos += n
            print(f"cpu oracle {name:16s} block={block:6d}  {total / (time.perf_counter() - t0) / 1e6:9.2f} Msamples/s", flush=True)
