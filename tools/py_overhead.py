#!/usr/bin/env python3
"""Where does the host time of a small-block render go?  (GPU box; cProfile over a 1024-frame block loop.)
argv[1]: c1 (GainPE(SinePE), pure) | c2 (BiquadPE(SinePE), stateful; default) | autowah (6 PEs, stateful)"""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygmu2_amd as pg
from pygmu2_amd import device
pg.set_sample_rate(44100)
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
if which == "c1":
    pe = pg.GainPE(pg.SinePE(440.0, 1.0, 0.0, channels=2), gain=0.5)
elif which == "autowah":
    from pygmu2_amd import transforms as tf
    src = pg.SinePE(frequency=220.0, amplitude=0.8)
    env = pg.EnvelopePE(src, attack=0.005, release=0.05, mode=pg.DetectionMode.PEAK)
    ctl = pg.TransformPE(env, func=tf.Chain(tf.Clip(0.0, 1.0), tf.Sqrt(), tf.Affine(2900.0, 100.0)), name="env_to_freq")
    pe = pg.GainPE(pg.BiquadPE(src, frequency=ctl, q=10.0, mode=pg.BiquadMode.LOWPASS), gain=1.0)
else:
    pe = pg.BiquadPE(pg.SinePE(440.0), 1000.0, 0.707)
r = pg.NullRenderer(44100); r.set_source(pe); r.start()
def loop(nblk, base=0):
    keep = None
    for i in range(nblk):
        keep = pe.render((base + i) * 1024, 1024)
    device.synchronize()
    return base + nblk
pos = loop(500)
t0 = time.perf_counter(); pos = loop(4000, pos); dt = time.perf_counter() - t0
print(f"{which}: {dt / 4000 * 1e6:.2f} us per 1024-frame block = {1024 * 4000 / dt / 1e6:.1f} Msamples/s")
pr = cProfile.Profile(); pr.enable(); pos = loop(4000, pos); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
