#!/usr/bin/env python3
"""Timing probe for the ADSR kernel under different gate patterns (GPU box only)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygmu2_amd import device
from oracle import pe_oracle as O

lib = device.ensure_init()
K, n, sr = int(os.environ.get('PGX_ADSR_K', '512')), 48000, 48000.0
rec = np.zeros(K, dtype=device.ADSR_PARAMS)
a, d, r = O.adsr_slopes(0.01, 0.1, 0.7, 0.2, sr)
rec[:] = (a, d, r, 0.7, 0)
params = device.DeviceBuffer.from_host(rec)
out = device.DeviceBuffer((K, n), np.float32)
ws = device.DeviceBuffer((lib.pgx_adsr_workspace_bytes(K, n),), np.uint8)

def run(name, gate=None, gates=None, reps=20):
    state = device.DeviceBuffer((K, 3), np.float64, zero=True)
    if gate is not None:
        g = device.DeviceBuffer.from_host(gate)
    def launch(i):
        if gate is not None:
            device.check(lib.pgx_adsr_gated(out.ptr, n, g.ptr, n, K, n, params.ptr, state.ptr, ws.ptr))
        else:
            device.check(lib.pgx_adsr_gated_periodic(out.ptr, n, K, i * n, n, gates.ptr, params.ptr, state.ptr, ws.ptr, 0))
    for i in range(3):
        launch(i)
    e0, e1 = device.Event(), device.Event()
    e0.record()
    for i in range(reps):
        launch(3 + i)
    e1.record()
    print(f"{name:40s} {e1.elapsed_ms_since(e0) / reps * 1e3:9.1f} us")

run("gate all zero (idle, pure fast path)", gate=np.zeros((K, n), np.float32))
run("gate all one (sustain after 1st block)", gate=np.ones((K, n), np.float32))
g = np.zeros((K, n), np.float32); g[:, ::2000] = 1.0
run("24 one-sample pulses per block", gate=g)
g = np.zeros((K, n), np.float32)
for k in range(K):
    per = int(sr / (2.0 + 0.01 * k)); 
    idx = np.arange(n) % per
    g[k] = (idx < per // 2)
run("2 Hz square (loaded)", gate=g)
gp = np.zeros(K, dtype=device.GATE_PARAMS)
for k in range(K):
    gp[k] = ((2.0 + 0.01 * k) / sr, 0.0, 0.5)
run("2 Hz square (in-kernel PeriodicGate)", gates=device.DeviceBuffer.from_host(gp))
gp["dt"] = 0.0
run("in-kernel gate, dt=0 (always high)", gates=device.DeviceBuffer.from_host(gp))
