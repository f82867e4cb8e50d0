#!/usr/bin/env python3
"""
The reference's own throughput suite, benchmarks/benchmark_pes.py (:149-196 protocol, :257-383 configs),
for every config whose PEs exist here: 5 warm-up + 50 timed contiguous renders of 44 100 frames through a
started NullRenderer graph; Msamples/s = frames per render / mean render time.

Three device figures per config: "sync" waits for the stream after every render (the latency a caller that
reads each block sees), "pipelined" waits once after the 50 renders (the offline rate), and the pipelined rate
with read-ahead / look-ahead switched off (every render its own launch sequence).  With `cpu` the
oracle's restatement of the reference is timed on this host for the same graph, same protocol, 1 thread.

    python tools/bench_suite.py [cpu] > profiles/<name>.md
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle.golden_cases import S

SR, N, WARM, RUNS = 44100, 44100, 5, 50
sine = lambda f=440.0, a=1.0: S("SinePE", frequency=f, amplitude=a)
CONFIGS = [
    ("SinePE (440 Hz)", sine()),
    ("ConstantPE", S("ConstantPE", value=0.5)),
    ("PiecewisePE", S("PiecewisePE", points=[[0, 0.0], [44100, 1.0]])),
    ("IdentityPE", S("IdentityPE")),
    ("DiracPE", S("DiracPE")),
    ("BlitSawPE (440 Hz, auto M)", S("BlitSawPE", frequency=440.0)),
    ("BlitSawPE (440 Hz, M=20)", S("BlitSawPE", frequency=440.0, m=20)),
    ("SuperSawPE (7 voices)", S("SuperSawPE", frequency=440.0, voices=7)),
    ("SuperSawPE (3 voices)", S("SuperSawPE", frequency=440.0, voices=3)),
    ("GainPE (constant)", S("GainPE", source=sine(), gain=0.5)),
    ("GainPE (modulated)", S("GainPE", source=sine(), gain=sine(5.0))),
    ("DelayPE (1000 samples)", S("DelayPE", source=sine(), delay=1000)),
    ("CropPE", S("CropPE", source=sine(), start=0, duration=44100)),
    ("MixPE (2 sources)", S("MixPE", inputs=[sine(440.0), sine(550.0)])),
    ("MixPE (4 sources)", S("MixPE", inputs=[sine(440.0), sine(550.0), sine(660.0), sine(880.0)])),
    ("LoopPE", S("LoopPE", source=S("CropPE", source=sine(), start=0, duration=4410))),
    ("DynamicsPE (compress)", S("DynamicsPE", source=sine(), envelope=S("EnvelopePE", source=sine()), mode="compress",
                                threshold=-10.0, ratio=4.0)),
    ("CompressorPE", S("CompressorPE", source=sine())),
    ("LimiterPE", S("LimiterPE", source=sine())),
    ("ExpanderPE", S("ExpanderPE", source=sine())),
    ("EnvelopePE", S("EnvelopePE", source=sine())),
    ("WindowPE (max)", S("WindowPE", source=sine())),
    ("BiquadPE (lowpass, fixed)", S("BiquadPE", source=sine(), mode="lowpass", frequency=1000.0, q=0.707)),
    ("BiquadPE (bandpass, fixed)", S("BiquadPE", source=sine(), mode="bandpass", frequency=1000.0, q=2.0)),
    ("BiquadPE (lowpass, modulated freq)", S("BiquadPE", source=sine(), mode="lowpass",
                                            frequency=sine(5.0, 500.0), q=0.707)),
    ("BiquadPE (bandpass, modulated Q)", S("BiquadPE", source=sine(), mode="bandpass", frequency=1000.0,
                                          q=sine(2.0, 1.0))),
    ("SVFilterPE (lowpass, fixed)", S("SVFilterPE", source=sine(), mode="lowpass", frequency=1000.0, q=0.707)),
    ("SVFilterPE (bandpass, fixed)", S("SVFilterPE", source=sine(), mode="bandpass", frequency=1000.0, q=2.0)),
    ("SVFilterPE (lowpass, modulated freq)", S("SVFilterPE", source=sine(), mode="lowpass",
                                              frequency=sine(5.0, 500.0), q=0.707)),
]
MISSING = ["RandomPE x3 (random_pe.py is disabled in the reference: the import fails there too)"]


def device_rates(spec, modes=("sync", "pipelined", "block_by_block")):
    """Msamples/s of one config under the reference's protocol.
    sync: wait for the device after every render; pipelined: wait once after the 50 renders; block_by_block:
    pipelined with read-ahead / look-ahead switched off (every render is its own launch sequence).
    The timed renders start away from the warm-up renders, so every frame of the timed region is rendered
    inside it (a look-ahead window opened during the warm-up is never served from); the last warm-up render is the
    block just before them."""
    import pygmu2_amd as pg
    from pygmu2_amd import device, look_ahead, read_ahead
    import spec_build
    pg.set_sample_rate(SR)
    out = {}
    for mode in modes:
        ahead = mode != "block_by_block"
        look_ahead.set_enabled(ahead)
        read_ahead.set_enabled(ahead)
        try:
            pe = spec_build.build(spec)
            r = pg.NullRenderer(sample_rate=SR)
            r.set_source(pe)
            r.start()
            first = (WARM + 1000) * N
            for i in range(WARM - 1):
                keep = pe.render(i * N, N)
            keep = pe.render(first - N, N)          # the seek (old window settled, one render outside any window)
            device.synchronize()                    # belongs to the warm-up: the timed renders are one stream
            if mode == "sync":
                times = []
                for i in range(RUNS):
                    t0 = time.perf_counter()
                    keep = pe.render(first + i * N, N)
                    device.synchronize()
                    times.append(time.perf_counter() - t0)
                mean = float(np.mean(times))
            else:
                t0 = time.perf_counter()
                for i in range(RUNS):
                    keep = pe.render(first + i * N, N)
                keep.dev
                device.synchronize()
                mean = (time.perf_counter() - t0) / RUNS
            r.stop()
        finally:
            look_ahead.set_enabled(True)
            read_ahead.set_enabled(True)
        out[mode] = N / mean / 1e6
    return out


def cpu_rate(spec, budget_s=4.0):
    from oracle import graph_eval
    g = graph_eval.Node(spec, SR)
    for i in range(2):
        g.render(i * N, N)
    times = []
    t_all = 0.0
    i = 2
    while len(times) < RUNS and t_all < budget_s:
        t0 = time.perf_counter()
        g.render(i * N, N)
        dt = time.perf_counter() - t0
        times.append(dt)
        t_all += dt
        i += 1
    return N / float(np.mean(times)) / 1e6


def main():
    with_cpu = "cpu" in sys.argv[1:]
    print("# benchmark_pes.py suite on MI355X (44 100-frame renders, 5 warm-up + 50 timed, Msamples/s)\n")
    print("Protocol and configs: reference `benchmarks/benchmark_pes.py:149-196, 257-383`; made by `tools/bench_suite.py`.\n")
    head = "| config | device, sync per render | device, pipelined | pipelined, look-ahead off |"
    sep = "|---|---|---|---|"
    if with_cpu:
        head += " CPU oracle, 1 thread | pipelined / CPU |"
        sep += "---|---|"
    print(head)
    print(sep)
    for name, spec in CONFIGS:
        rates = device_rates(spec)
        s, p = rates["sync"], rates["pipelined"]
        row = f"| {name} | {s:.0f} | {p:.0f} | {rates['block_by_block']:.0f} |"
        if with_cpu:
            c = cpu_rate(spec)
            row += f" {c:.1f} | {p / c:.0f}x |"
        print(row, flush=True)
    print(f"\nNot built: {', '.join(MISSING)}.")
    if with_cpu:
        print("CPU oracle = oracle/ (numpy/scipy + the C restatement of the numba kernels); its SVF coefficient "
              "loop and SuperSaw are plain Python/numpy, so those rows understate a numba-equipped reference.")


if __name__ == "__main__":
    main()
