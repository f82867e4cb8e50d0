#!/usr/bin/env python3
"""Host cost of the pieces of a C2 look-ahead window opening (GPU box), each timed alone over many repetitions, and of
the opening as look_ahead.render does it."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import pygmu2_amd as pg
from pygmu2_amd import device, look_ahead
from pygmu2_amd._kernels import new_output, lib, ptr
from pygmu2_amd.snippet import Snippet

pe, r = bench.c2_graph(pg)
frames = 1_000_000
for i in range(3):
    pe.render(i * frames, frames)
device.synchronize()
L = lib()
REP = 400


def timed(name, fn, rep=REP, sync_every=50):
    t_all = 0.0
    for i in range(rep):
        t0 = time.perf_counter()
        fn()
        t_all += time.perf_counter() - t0
        if i % sync_every == sync_every - 1:
            device.synchronize()
    print(f"{name:58s} {t_all / rep * 1e6:7.2f} us")


small = frames * 8
timed("DeviceBuffer((2,), f64)  (snapshot target)", lambda: device.DeviceBuffer((1, 2), np.float64))
timed("new_output(8 M frames, 1)", lambda: new_output(small, 1))
out = new_output(small, 1)
w, amp, phase = pe._sine_chain()
sr = float(pe.sample_rate)
bk = device.DeviceBuffer((1, 2), np.float64)
pos = [10 ** 9]


def launch():
    L.pgx_biquad_sine(out.ptr, pos[0], small, sr, w, amp, phase, pe._coef.ptr, pe._tables.ptr, pe._settle,
                      pe._state.ptr, bk.ptr)
    pos[0] += small


timed("pgx_biquad_sine ctypes call (8 M frames)", launch, rep=200, sync_every=10)
timed("pgx_stream_is_forked (a trivial ctypes call)", lambda: L.pgx_stream_is_forked())
timed("Snippet(start, DeviceBuffer)", lambda: Snippet(0, out))
timed("Snippet.window_rows", lambda: Snippet.window_rows(0, out, 0, frames))
timed("take_snapshot(nodes)", lambda: [n._flush_backup() for n, _ in look_ahead.take_snapshot(pe.__dict__["_la_nodes"])
                                       if hasattr(n, "_flush_backup")])
timed("pe._render(start, 8 M) (fused launch + buffers)", lambda: (pe._render(pos[0], small), pos.__setitem__(0, pos[0] + small)),
      rep=200, sync_every=10)

# the opening itself: windows of 2 blocks, every second pull opens one
look_ahead.AHEAD_BLOCKS = 2
p = [0]
pe2, r2 = bench.c2_graph(pg)
for i in range(4):
    pe2.render(i * frames, frames)
p[0] = 4 * frames
opens, serves = [], []
for i in range(400):
    t0 = time.perf_counter()
    pe2.render(p[0], frames)
    dt = time.perf_counter() - t0
    (serves if pe2.__dict__["_la_win"].served != p[0] + frames or dt < 3e-6 else opens).append(dt)
    p[0] += frames
    if i % 20 == 19:
        device.synchronize()
opens.sort(); serves.sort()
print(f"window opening (2 x 1 M frames): median {opens[len(opens) // 2] * 1e6:.2f} us over {len(opens)}; "
      f"serving a row: median {serves[len(serves) // 2] * 1e6:.2f} us over {len(serves)}")
