#!/usr/bin/env python3
"""
bench.py -- throughput of the MI355X render path on the BASELINE.json configurations.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c1|c3|c4|c5]

Prints ONE JSON line (rank 0).  Metric: Msamples/s = output frames rendered per wall second
x 1e-6, the metric of the reference's benchmarks/benchmark_pes.py:62-66.

Primary workload (BASELINE.json configs[1], "C2"): BiquadPE(SinePE(440), 1000 Hz, q .707,
LOWPASS), 44.1 kHz mono, one step = one contiguous render(start, 1_000_000) through the
public PE API on a started NullRenderer graph, inputs generated on the device (SinePE),
outputs left in HBM.  K steps are timed between barrier+synchronize pairs.
With --gpus N > 1 every rank renders its own replica of the chain (a single biquad chain is
one sequence: "replicas only"); the sharded 512-voice mix with its RCCL reduction is
reported alongside in the `voice_mix` object (strong scaling).

Extra objects in the same JSON line:
  roofline      dominant kernel pair of C2 (k_biquad_const reduce+apply), algorithmic bytes
                8 B/frame, timed live with HIP events on the library stream
  roofline_scaled  the same entry point on 2^26 frames (past launch-latency / cache effects)
  cpu_baseline  the CPU oracle (numpy sin + scipy lfilter, i.e. the reference's own
                primitives in the reference's order) on the same workload, 1 thread
  cases         other BASELINE configs measured in the same run (C1 sine->gain blocks,
                C3 convolution) with their own CPU-oracle timings
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c4", "c5"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU oracle timings")
    ap.add_argument("--no-extras", action="store_true", help="primary workload only")
    return ap.parse_args()


# ----------------------------------------------------------------------------- distributed glue
class Dist:
    def __init__(self, n_gpus):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.enabled = self.world > 1
        self.torch = None
        if self.enabled:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            torch.cuda.set_device(self.local_rank)
            dist.init_process_group(backend="nccl")      # RCCL on ROCm

    def barrier(self):
        if self.enabled:
            self.dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if not self.enabled:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def shutdown(self):
        if self.enabled:
            self.dist.destroy_process_group()


def timed_steps(dist: Dist, step, steps: int, warmup: int):
    """W untimed + K timed calls of step(i); barrier + device sync on both sides; max over ranks."""
    from pygmu2_amd import device
    for i in range(warmup):
        step(i)
    device.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    device.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    return dist.max_over_ranks(dt)


# ----------------------------------------------------------------------------- workloads
def c2_graph(pg):
    pg.set_sample_rate(44100)
    pe = pg.BiquadPE(pg.SinePE(frequency=440.0), frequency=1000.0, q=0.707, mode=pg.BiquadMode.LOWPASS)
    r = pg.NullRenderer(sample_rate=44100)
    r.set_source(pe)
    r.start()
    return pe, r


def bench_c2(pg, dist, steps, warmup, frames=1_000_000):
    pe, r = c2_graph(pg)
    keep = {}

    def step(i):
        keep["s"] = pe.render(i * frames, frames)      # stays in HBM

    dt = timed_steps(dist, step, steps, warmup)
    r.stop()
    return dt, frames


def pmc_traffic(entry: str, frames: int):
    """HBM bytes per launch from the committed PMC measurement (profiles/traffic.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(entry, {}).get(str(frames))
    except (OSError, ValueError):
        return None


def biquad_kernel_roofline(pg, frames, launches, settled=True):
    """HIP-event timing of the pgx_biquad_const entry point alone (input resident in HBM).

    settled=True is what BiquadPE passes for the C2 section (one k_biquad_settled launch);
    settled=False times the exact reduce + apply pair that slowly decaying sections use."""
    from pygmu2_amd import device
    from pygmu2_amd.biquad_pe import rbj_coefficients, settle_frames
    lib = device.ensure_init()
    pg.set_sample_rate(44100)
    x = pg.SinePE(frequency=440.0).render(0, frames).dev
    out = device.DeviceBuffer((frames, 1), np.float32)
    c = rbj_coefficients(pg.BiquadMode.LOWPASS, 1000.0, 0.707, 0.0, 44100.0)
    coef = device.DeviceBuffer.from_host(np.asarray(c, dtype=np.float64))
    settle = settle_frames(c[3], c[4]) if settled else 0
    state = device.DeviceBuffer((1, 2), np.float64, zero=True)
    tables = device.DeviceBuffer((lib.pgx_biquad_table_doubles(),), np.float64)
    device.check(lib.pgx_biquad_tables(tables.ptr, coef.ptr, 1))
    need = lib.pgx_biquad_workspace_bytes(1, frames, 1, settle)
    ws = device.DeviceBuffer((max(need, 1),), np.uint8)
    kernel = ("k_biquad_const<reduce>+<apply> (pgx_biquad_const, settle_frames=0)" if need else
              f"k_biquad_settled (pgx_biquad_const, settle_frames={settle})")

    def launch():
        device.check(lib.pgx_biquad_const(out.ptr, 0, x.ptr, 0, 1, frames, 1, coef.ptr, tables.ptr if settle else None, settle, state.ptr,
                                          ws.ptr))

    for _ in range(3):
        launch()
    e0, e1 = device.Event(), device.Event()
    e0.record()
    for _ in range(launches):
        launch()
    e1.record()
    ms = e1.elapsed_ms_since(e0) / launches
    algo_bytes = 8.0 * frames                      # read f32 + write f32 per frame
    achieved = algo_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": pmc_traffic("pgx_biquad_const" if need else "pgx_biquad_const_settled", frames),
            "kernel": kernel,
            "frames_per_launch": frames, "algorithmic_bytes_per_launch": algo_bytes,
            "avg_launch_ms": round(ms, 6)}


def cpu_c2(frames, budget_s=12.0):
    """CPU oracle: np.sin source + scipy lfilter biquad, single thread, same chain and size."""
    from oracle import pe_oracle as O
    st = O.biquad_state(1)
    t_all, reps, pos = 0.0, 0, 0
    while t_all < budget_s and reps < 5000:
        t0 = time.perf_counter()
        x = O.sine_pure(pos, frames, 440.0, sr=44100)
        O.biquad_const(st, x, 1000.0, 0.707, "lowpass", 0.0, 44100)
        t_all += time.perf_counter() - t0
        pos += frames
        reps += 1
    return {"value": round(frames * reps / t_all / 1e6, 3), "unit": "Msamples/s", "cores": 1,
            "kind": "port",
            "sample": f"{reps} x render of {frames} frames: oracle sine_pure + biquad_const "
                      f"(numpy sin + scipy.signal.lfilter, float64), 1 thread, {t_all:.1f} s"}


def bench_c1(pg, dist, steps, warmup):
    """C1: GainPE(SinePE(440, ch=2), 0.5), 44.1 kHz, 431 blocks of 1024 (441 000 frames) per step."""
    pg.set_sample_rate(44100)
    pe = pg.GainPE(pg.SinePE(frequency=440.0, amplitude=1.0, phase=0.0, channels=2), gain=0.5)
    r = pg.NullRenderer(sample_rate=44100)
    r.set_source(pe)
    r.start()
    total = 441_000
    keep = {}

    def step(i):
        pos = 0
        while pos < total:
            n = min(1024, total - pos)
            keep["s"] = pe.render(pos, n)
            pos += n

    dt = timed_steps(dist, step, steps, warmup)
    r.stop()
    return dt, total


def hello_sine_case(pg):
    """examples/01_hello_sine.py:41-56, what BASELINE's config 0 cites: a C-major triad of three SinePEs
    (amplitude 0.3) -> MixPE -> GainPE(0.3) -> CropPE(8 s), pulled in 1024-frame blocks at 44.1 kHz."""
    pg.set_sample_rate(44100)
    total = 8 * 44100
    triad = [pg.SinePE(frequency=440.0 * 2.0 ** ((p - 69) / 12.0), amplitude=0.3) for p in (60, 64, 67)]
    root = pg.CropPE(pg.GainPE(pg.MixPE(*triad), gain=0.3), 0, total)
    r = pg.NullRenderer(sample_rate=44100)
    r.set_source(root)
    r.start()
    from pygmu2_amd import device
    best = 0.0
    for rep in range(4):
        device.synchronize()
        t0 = time.perf_counter()
        pos = 0
        while pos < total:
            n = min(1024, total - pos)
            keep = root.render(pos, n)
            pos += n
        device.synchronize()
        if rep:
            best = max(best, total / (time.perf_counter() - t0) / 1e6)
    r.stop()
    return round(best, 3)


def cpu_hello_sine(budget_s=4.0):
    from oracle import pe_oracle as O
    total, t_all, reps = 8 * 44100, 0.0, 0
    freqs = [440.0 * 2.0 ** ((p - 69) / 12.0) for p in (60, 64, 67)]
    while t_all < budget_s and reps < 50:
        t0 = time.perf_counter()
        pos = 0
        while pos < total:
            n = min(1024, total - pos)
            O.gain_const(O.mix([O.sine_pure(pos, n, f, 0.3, 0.0, 44100, 1) for f in freqs]), 0.3)
            pos += n
        t_all += time.perf_counter() - t0
        reps += 1
    return round(total * reps / t_all / 1e6, 3)


def cpu_c1(budget_s=5.0):
    from oracle import pe_oracle as O
    total, t_all, reps = 441_000, 0.0, 0
    while t_all < budget_s and reps < 100:
        t0 = time.perf_counter()
        pos = 0
        while pos < total:
            n = min(1024, total - pos)
            O.gain_const(O.sine_pure(pos, n, 440.0, 1.0, 0.0, 44100, 2), 0.5)
            pos += n
        t_all += time.perf_counter() - t0
        reps += 1
    return round(total * reps / t_all / 1e6, 3)


def autowah_case(pg, kind, block=1024, seconds=8):
    """The autowah graph of benchmarks/profile_biquad_vs_svfilter.py:47-72 (BASELINE config 2's script):
    source -> EnvelopePE -> TransformPE(env -> 100..3000 Hz) -> BiquadPE | SVFilterPE(frequency=PE, q=10) -> GainPE,
    cropped to 8 s and rendered in 1024-frame blocks through the Renderer, as the script does."""
    from pygmu2_amd import device, transforms as tf
    pg.set_sample_rate(44100)
    src = pg.SinePE(frequency=220.0, amplitude=0.8)
    env = pg.EnvelopePE(src, attack=0.005, release=0.05, mode=pg.DetectionMode.PEAK)
    ctl = pg.TransformPE(env, func=tf.Chain(tf.Clip(0.0, 1.0), tf.Sqrt(), tf.Affine(2900.0, 100.0)),
                         name="env_to_freq")
    flt = (pg.BiquadPE if kind == "biquad" else pg.SVFilterPE)(src, frequency=ctl, q=10.0,
                                                               mode=pg.BiquadMode.LOWPASS)
    total = 44100 * seconds
    root = pg.CropPE(pg.GainPE(flt, gain=1.0), 0, total)
    r = pg.NullRenderer(sample_rate=44100)
    r.set_source(root)
    best = None
    for rep in range(3):                              # first pass warms allocations; keep the best of the rest
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
        if rep and (best is None or dt < best):
            best = dt
    return round(total / best / 1e6, 3)


def cpu_autowah(kind, block=1024, seconds=8):
    from oracle import graph_eval
    from oracle.golden_cases import S
    src = S("SinePE", frequency=220.0, amplitude=0.8)
    env = S("EnvelopePE", source=src, attack=0.005, release=0.05, mode="peak")
    ctl = S("TransformPE", source=env, ops=[["clip", 0.0, 1.0], ["sqrt"], ["affine", 2900.0, 100.0]])
    g = graph_eval.Node(S("GainPE", source=S("BiquadPE" if kind == "biquad" else "SVFilterPE", source=src,
                                             frequency=ctl, q=10.0, mode="lowpass"), gain=1.0), 44100)
    total = 44100 * seconds
    t0 = time.perf_counter()
    pos = 0
    while pos < total:
        n = min(block, total - pos)
        g.render(pos, n)
        pos += n
    return round(total / (time.perf_counter() - t0) / 1e6, 3)


def c3_inputs(frames):
    x = (np.random.default_rng(0).standard_normal((frames, 2)) * 0.1).astype(np.float32)
    n = np.arange(65536)
    h = (np.random.default_rng(1).standard_normal(65536) * np.exp(-n / 8000.0)).astype(np.float32)
    return x, h


def bench_c3(pg, dist, steps, warmup, frames=96_000):
    """C3: ConvolvePE(stereo ArrayPE, 65 536-tap FIR, fft_size=131072), 48 kHz; a step renders the
    whole `frames`-long signal in one call on a fresh (history-cleared) stream position."""
    pg.set_sample_rate(48000)
    x, h = c3_inputs(frames)
    pe = pg.ConvolvePE(pg.ArrayPE(x), pg.ArrayPE(h), fft_size=131072)
    r = pg.NullRenderer(sample_rate=48000)
    r.set_source(pe)
    r.start()
    keep = {}

    def step(i):
        keep["s"] = pe.render(0, frames)

    dt = timed_steps(dist, step, steps, warmup)
    r.stop()
    return dt, frames


def conv_kernel_roofline(pg, frames, launches):
    from pygmu2_amd import device
    lib = device.ensure_init()
    x, h = c3_inputs(frames)
    xd, hd = device.DeviceBuffer.from_host(x), device.DeviceBuffer.from_host(h.reshape(-1, 1))
    out = device.DeviceBuffer((frames, 2), np.float32)
    hist = device.DeviceBuffer((65535, 2), np.float32, zero=True)
    ws = device.DeviceBuffer((lib.pgx_convolve_workspace_bytes(frames, 65536, 2),), np.uint8)

    def launch():
        device.check(lib.pgx_convolve(out.ptr, xd.ptr, frames, 2, hd.ptr, 65536, 1, 2, hist.ptr, ws.ptr))

    for _ in range(2):
        launch()
    e0, e1 = device.Event(), device.Event()
    e0.record()
    for _ in range(launches):
        launch()
    e1.record()
    ms = e1.elapsed_ms_since(e0) / launches
    flops = 2.0 * 65536 * 2 * frames                # direct form: 2*L*C_out per frame
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(achieved, 3), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 5), "traffic": None,
            "kernel": "k_conv_mfma<4096> (+prep/reduce/hist, pgx_convolve)",
            "frames_per_launch": frames, "algorithmic_flops_per_launch": flops, "avg_launch_ms": round(ms, 6)}


def conv_fft_roofline(pg, frames, launches):
    """HIP-event timing of pgx_convolve_fft (the path ConvolvePE takes for the 65 536-tap C3 filter)."""
    from pygmu2_amd import device
    lib = device.ensure_init()
    x, h = c3_inputs(frames)
    L = 65536
    nfft = lib.pgx_convolve_fft_size(L)
    xd, hd = device.DeviceBuffer.from_host(x), device.DeviceBuffer.from_host(h.reshape(-1, 1))
    spec = device.DeviceBuffer((lib.pgx_convolve_fft_spectrum_bytes(nfft, 1),), np.uint8)
    device.check(lib.pgx_convolve_fft_prepare(spec.ptr, hd.ptr, L, 1, nfft))
    out = device.DeviceBuffer((frames, 2), np.float32)
    hist = device.DeviceBuffer((L - 1, 2), np.float32, zero=True)
    ws = device.DeviceBuffer((lib.pgx_convolve_fft_workspace_bytes(frames, L, 2, nfft),), np.uint8)

    def launch():
        device.check(lib.pgx_convolve_fft(out.ptr, xd.ptr, frames, 2, spec.ptr, L, 1, 2, nfft, hist.ptr, ws.ptr))

    for _ in range(2):
        launch()
    e0, e1 = device.Event(), device.Event()
    e0.record()
    for _ in range(launches):
        launch()
    e1.record()
    ms = e1.elapsed_ms_since(e0) / launches
    algo_bytes = 4.0 * (2 + 2) * frames              # 4(C_in + C_out) per frame; taps are read once
    hop = nfft - (L - 1)
    pairs = 2 * ((-(-frames // hop) + 1) // 2)
    moved = 3.0 * 32.0 * nfft * pairs                # three passes, 16 B read + 16 B written per complex point
    achieved = algo_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
            "work_buffer_bytes_per_launch": moved,
            "kernel": "k_fft_cols<fwd> + k_fft_rows + k_fft_cols<inv> + k_fft_hist (pgx_convolve_fft, "
                      f"N={nfft}, float64)",
            "frames_per_launch": frames, "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(ms, 6)}


def cpu_c3(frames=96_000, budget_s=6.0):
    from oracle import pe_oracle as O
    x, h = c3_inputs(frames)
    t_all, reps = 0.0, 0
    while t_all < budget_s and reps < 50:
        st = O.convolve_state()
        t0 = time.perf_counter()
        O.convolve(st, 0, x, h, fft_size=131072)
        t_all += time.perf_counter() - t0
        reps += 1
    return round(frames * reps / t_all / 1e6, 3)


# ----------------------------------------------------------------------------- main
def main():
    args = parse_args()
    dist = Dist(args.gpus)
    if dist.enabled:
        os.environ.setdefault("PYGMU_DEVICE", str(dist.local_rank))
    import pygmu2_amd as pg
    from pygmu2_amd import device
    device.ensure_init()

    n_gpus = max(1, dist.world)
    result = {}
    if args.workload == "c2":
        dt, frames = bench_c2(pg, dist, args.steps, args.warmup)
        name, sr = "C2: BiquadPE(SinePE(440), lowpass 1 kHz, q 0.707), 44.1 kHz mono, render(start, 1_000_000) per step", 44100
        units = frames * args.steps * n_gpus
    elif args.workload == "c1":
        dt, frames = bench_c1(pg, dist, args.steps, args.warmup)
        name = "C1: GainPE(SinePE(440, ch=2), 0.5), 44.1 kHz stereo, 431 blocks of 1024 per step"
        units = frames * args.steps * n_gpus
    elif args.workload == "c3":
        dt, frames = bench_c3(pg, dist, args.steps, args.warmup)
        name = "C3: ConvolvePE stereo x 65536-tap FIR, 48 kHz, 96 000 frames per step"
        units = frames * args.steps * n_gpus
    elif args.workload == "c4":
        from pygmu2_amd.sharding import bench_voice_mix
        dt, frames, name = bench_voice_mix(pg, dist, args.steps, args.warmup, voices=64, config="c4")
        units = frames * args.steps
    else:
        from pygmu2_amd.sharding import bench_voice_mix
        dt, frames, name = bench_voice_mix(pg, dist, args.steps, args.warmup)
        units = frames * args.steps

    value = units / dt / 1e6
    result.update({
        "metric": "Msamples/s rendered (benchmark_pes.py metric: output frames / wall second)",
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 6), "higher_is_better": True,
        "scaling": "strong" if args.workload in ("c4", "c5") else "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": name, "frames_per_step": frames,
                   "parallelism": ("voices sharded over ranks, RCCL all-reduce of the partial mixes"
                                   if args.workload in ("c4", "c5") else
                                   ("single chain" if n_gpus == 1 else f"{n_gpus} independent replicas (replicas only)"))},
    })

    if args.workload not in ("c4", "c5") and not args.no_extras:
        # collective: every rank takes part.  512-voice mix sharded over the ranks (strong scaling).
        from pygmu2_amd.sharding import bench_voice_mix
        vdt, vframes, vname = bench_voice_mix(pg, dist, 10, 2)
        result["voice_mix"] = {"value": round(vframes * 10 / vdt / 1e6, 3), "unit": "Msamples/s",
                               "voice_msamples_s": round(512 * vframes * 10 / vdt / 1e6, 1),
                               "ms_per_block": round(vdt / 10 * 1e3, 4), "scaling": "strong",
                               "workload": vname, "steps": 10, "warmup": 2}
        # north_star's scaling case: 512 SuperSaw voices (3584 oscillators) -- throughput-bound, so it is the
        # sharded workload that can scale with the GPU count (the C5 voices above are latency-bound chains)
        sdt, sframes, sname = bench_voice_mix(pg, dist, 6, 2, config="supersaw")
        result["supersaw_mix"] = {"value": round(sframes * 6 / sdt / 1e6, 3), "unit": "Msamples/s",
                                  "oscillator_msamples_s": round(3584 * sframes * 6 / sdt / 1e6, 1),
                                  "ms_per_block": round(sdt / 6 * 1e3, 4), "scaling": "strong",
                                  "workload": sname, "steps": 6, "warmup": 2}

    if dist.rank == 0 and not args.no_extras:
        result["device"] = device.device_name()
        result["roofline"] = biquad_kernel_roofline(pg, 1_000_000, 200)
        result["roofline_scaled"] = biquad_kernel_roofline(pg, 1 << 26, 10)
        cases = {}
        if args.workload == "c2" and n_gpus == 1:
            dt1, f1 = bench_c1(pg, Dist(1), 5, 1)
            cases["c1_sine_gain_1024_blocks"] = {"value": round(f1 * 5 / dt1 / 1e6, 3), "unit": "Msamples/s"}
            cases["c1_hello_sine_example_1024_blocks"] = {"value": hello_sine_case(pg), "unit": "Msamples/s"}
            dt3, f3 = bench_c3(pg, Dist(1), 10, 2)
            cases["c3_convolve_64k_taps"] = {"value": round(f3 * 10 / dt3 / 1e6, 3), "unit": "Msamples/s",
                                             "path": "float64 FFT overlap-save (pgx_convolve_fft)",
                                             "roofline": conv_fft_roofline(pg, 96_000, 20),
                                             # the dense FIR x block product on the matrix cores, same filter:
                                             # what ConvolvePE uses below convolve_pe.FFT_MIN_TAPS taps
                                             "direct_form_mfma": conv_kernel_roofline(pg, 96_000, 10)}
        if args.workload == "c2" and n_gpus == 1:
            from pygmu2_amd.sharding import bench_voice_mix
            d4, f4, _ = bench_voice_mix(pg, Dist(1), 5, 1, voices=64, config="c4")
            cases["c4_supersaw_ladder_mix_64"] = {"value": round(f4 * 5 / d4 / 1e6, 3), "unit": "Msamples/s",
                                                  "ms_per_block": round(d4 / 5 * 1e3, 4)}
            cases["autowah_biquad_1024_blocks"] = {"value": autowah_case(pg, "biquad"), "unit": "Msamples/s"}
            cases["autowah_svf_1024_blocks"] = {"value": autowah_case(pg, "svf"), "unit": "Msamples/s"}
        if not args.no_cpu and n_gpus == 1:
            result["cpu_baseline"] = cpu_c2(1_000_000)
            if "autowah_biquad_1024_blocks" in cases:
                # the oracle's varying biquad is the C restatement of the numba kernel; its SVF coefficient
                # loop is plain Python (slow), so only the biquad graph gets a CPU figure
                cases["autowah_biquad_1024_blocks"]["cpu_oracle_msamples_s"] = cpu_autowah("biquad")
            if "c1_sine_gain_1024_blocks" in cases:
                cases["c1_sine_gain_1024_blocks"]["cpu_oracle_msamples_s"] = cpu_c1()
                cases["c1_hello_sine_example_1024_blocks"]["cpu_oracle_msamples_s"] = cpu_hello_sine()
            if "c3_convolve_64k_taps" in cases:
                cases["c3_convolve_64k_taps"]["cpu_oracle_msamples_s"] = cpu_c3()
        if cases:
            result["cases"] = cases

    if dist.rank == 0:
        print(json.dumps(result), flush=True)
    dist.shutdown()


if __name__ == "__main__":
    main()
