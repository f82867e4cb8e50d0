#!/usr/bin/env python3
"""
bench.py -- throughput of the MI355X render path on the BASELINE.json configurations.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c1|c3|c4|c5|supersaw]

Prints ONE JSON line (rank 0).  Metric: Msamples/s = output frames rendered per wall second x 1e-6, the
metric of the reference's benchmarks/benchmark_pes.py:62-66.

Launching.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process only spawns N fresh
rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, one per GPU), relays rank 0's
JSON line and exits non-zero if any rank did; it never touches the GPU itself.  Under a launcher that
already set WORLD_SIZE (python -m torch.distributed.run ...) the process is one rank.  Ranks meet through
the launcher's c10d store (128-byte communicator id), then talk RCCL through the library's own entry points
(pgx_comm_init / pgx_allreduce_sum); the barrier and the max-over-ranks of the timed region are scalar
all-reduces on the same communicator.  `--dry-run` exercises exactly this glue on CPU-only hosts (no
library call, no measurement): every rank reports through the store and the line carries n_gpus /
n_ranks_seen / the shard sizes.

Primary workload (BASELINE.json configs[1], "C2"): BiquadPE(SinePE(440), 1000 Hz, q .707, LOWPASS), 44.1 kHz
mono, one step = one contiguous render(start, 1_000_000) through the public PE API on a started
NullRenderer graph, inputs generated on the device, outputs left in HBM.  With --gpus N every rank renders
its own replica of the chain (a single biquad chain is one sequence: "replicas only").  The workloads that
shard (c4, c5, supersaw: inputs of the root MixPE dealt i mod N over the ranks, one RCCL all-reduce of the
partial mix per block) are selected with --workload and are reported inside the default line as well.

Extra objects in the default line (rank 0, N = 1): roofline / roofline_scaled (the C2 filter kernel, HIP
events), cpu_baseline (the CPU oracle on this host, 1 thread, host core count stated), value_with_d2h
(the same C2 steps with every root Snippet read on the host: pinned buffer, async copy overlapped with the
next step -- and with a sync per step), cases (the other BASELINE configs, each with its CPU figure), suite
(benchmark_pes.py's own protocol: 44 100-frame renders, 5 + 50, sync and pipelined, CPU beside).
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
F64_PEAK_TSLOTS = 256 * 4 * 16 * 2.4e9 / 1e12   # float64 lane-slots/s: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T
                                                # (78.6 TFLOP/s with an FMA counted as two: SURVEY Appendix B)
SHARDED = ("c4", "c5", "supersaw")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (the defaults are a stream: 20 000 steps of C2 are 25 ms of device time -- look-ahead windows at their full size, the chip
    # at the clock it holds; 200 steps after 20, the defaults of rounds 1 - 4, are a burst of 0.3 ms: 680 000 against 856 000)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c4", "c5", "supersaw"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU oracle timings")
    ap.add_argument("--no-extras", action="store_true", help="primary workload only")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch glue only (CPU host): ranks rendezvous and report, nothing is rendered")
    ap.add_argument("--allow-no-rccl", action="store_true",
                    help="N > 1 ranks without an RCCL communicator: go on with store barriers (n_ranks_seen = 0, "
                         "collective = 'store', sharded mixes reported as unavailable) instead of failing")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------- launching N ranks
SPAWN_DEADLINE = float(os.environ.get("PGX_BENCH_DEADLINE", "1500"))     # seconds for the whole N-rank run


def spawn_ranks(n: int) -> int:
    """Parent of a `--gpus N` run: N child ranks of this same command line; relay rank 0's stdout.  The whole run
    has a deadline: a rank stuck in a collective or at the store takes its siblings down with it instead of
    hanging the parent."""
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PGX_BENCH_SPAWNED="1")
        # This pool's host driver shares device memory between processes through dmabuf only: with the legacy IPC
        # mode RCCL's peer-to-peer setup fails in hipIpcGetMemHandle ("invalid argument").  The driver's launcher
        # exports the same setting; a value the caller chose is kept.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, cwd=os.getcwd()))
    out0 = procs[0].stdout

    def relay():
        for line in iter(out0.readline, b""):            # rank 0's JSON line (and nothing else) goes to stdout
            text = line.decode("utf-8", "replace")
            dest = sys.stdout if text.lstrip().startswith("{") else sys.stderr     # (a library's banner: stderr)
            dest.write(text)
            dest.flush()

    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    deadline = time.time() + SPAWN_DEADLINE
    failed = 0
    try:
        pending = list(procs)
        while pending and time.time() < deadline:
            for p in list(pending):
                rc = p.poll()
                if rc is not None:
                    pending.remove(p)
                    failed = failed or rc
            if failed:
                break                                     # one rank failed: the others cannot finish a collective
            time.sleep(0.02)
        if pending:
            if not failed:
                print(f"[bench] {len(pending)} of {n} ranks still running after {SPAWN_DEADLINE:.0f} s: killed",
                      file=sys.stderr, flush=True)
            failed = failed or 1
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
        reader.join(timeout=5)
    return 1 if failed else 0


# ----------------------------------------------------------------------------- the ranks' control plane
RCCL_INIT_TIMEOUT = float(os.environ.get("PGX_BENCH_RCCL_TIMEOUT", "180"))


class Dist:
    """World / rank from the launcher's environment.  world > 1: the c10d store the launcher provides
    carries the communicator id (and, in a dry run, the whole report); RCCL carries everything else."""

    def __init__(self, dry_run=False, allow_no_rccl=False):
        self.allow_no_rccl = allow_no_rccl
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.enabled = self.world > 1
        self.dry_run = dry_run
        self.store = None
        self._barriers = 0
        self.collective = "rccl"             # "store" after a failed communicator build (connect)
        self.collective_error = None
        if self.enabled:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            from torch.distributed import rendezvous
            self.store, _, _ = next(rendezvous("env://", rank=self.rank, world_size=self.world))

    def connect(self):
        """After device init: build the RCCL communicator (every rank).  Returns the ranks seen by an
        all-reduce of ones on it.  If the communicator cannot be built on every rank within RCCL_INIT_TIMEOUT
        seconds (a rank's error, or peers stuck in the bootstrap) all ranks agree through the store to go on
        without it: barrier and max-over-ranks then ride on the store, the replica workload is measured as
        usual and the sharded mixes, which need the collective, are reported as unavailable."""
        if not self.enabled:
            return 1
        if self.dry_run:
            return int(self._store_sum("ranks_seen", 1))
        import threading
        from pygmu2_amd import comm
        outcome = {}

        def build():
            try:
                if self.rank == 0:
                    self.store.set("pgx_comm_id", comm.unique_id())
                ident = self.store.get("pgx_comm_id")
                comm.init(self.rank, self.world, bytes(ident))
                outcome["seen"] = int(round(comm.reduce_scalar(1.0, "sum")))
            except Exception as e:                                  # noqa: BLE001 - reported, then agreed on
                outcome["error"] = f"{type(e).__name__}: {e}"

        # RCCL greets with a version banner on stdout: stdout carries the one JSON line, so the file descriptor
        # points at stderr while the communicator is built (nothing else prints meanwhile: the rank waits here)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            t = threading.Thread(target=build, daemon=True)
            t.start()
            t.join(RCCL_INIT_TIMEOUT)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        ok = "seen" in outcome and outcome["seen"] == self.world
        timed_out = not ok and "error" not in outcome and t.is_alive()
        if not ok and "error" not in outcome:
            outcome["error"] = "timed out" if t.is_alive() else f"saw {outcome.get('seen')} of {self.world} ranks"
        all_ok = self._store_sum("rccl_ok", 1 if ok else 0) == self.world
        if all_ok:
            return outcome["seen"]                    # the all-reduce of ones on the communicator, nothing else
        self.collective = "store"
        self.collective_error = outcome.get("error", "a peer could not build the communicator")
        any_timeout = self._store_sum("rccl_timeout", 1 if timed_out else 0) > 0
        if any_timeout or not self.allow_no_rccl:
            # no communicator, no multi-GPU measurement: every rank leaves with an error (a timed-out bootstrap also
            # leaves a thread inside ncclCommInitRank that could still publish a communicator later: never go on)
            print(f"[bench] rank {self.rank}: RCCL communicator unavailable ({self.collective_error}); "
                  f"--gpus {self.world} needs it (pass --allow-no-rccl to measure replicas over store barriers)",
                  file=sys.stderr, flush=True)
            if self.rank == 0:
                print(json.dumps({"metric": "Msamples/s rendered (benchmark_pes.py metric: output frames / wall second)",
                                  "value": None, "unit": "Msamples/s", "n_gpus": self.world, "n_ranks_seen": 0,
                                  "collective": None,
                                  "error": f"no RCCL communicator: {self.collective_error}"}), flush=True)
            self._leave_store()
            sys.stdout.flush()
            os._exit(3)
        print(f"[bench] rank {self.rank}: RCCL communicator unavailable ({self.collective_error}); "
              f"--allow-no-rccl: barriers go through the store, n_ranks_seen = 0", file=sys.stderr, flush=True)
        self.ranks_present = int(self._store_sum("ranks_present", 1))
        return 0                                      # RCCL saw nobody

    def _store_sum(self, key, value):
        self.store.add(key, int(value))
        self.store.add(key + "_n", 1)
        t0 = time.time()
        while int(self.store.add(key + "_n", 0)) < self.world:
            if time.time() - t0 > 300:
                raise RuntimeError(f"rank {self.rank}: peers never arrived at {key}")
            time.sleep(0.0005)
        return int(self.store.add(key, 0))

    def _store_max(self, key, value):
        """max of one float over the ranks through the store (microsecond resolution)."""
        self.store.set(f"{key}_r{self.rank}", repr(float(value)))
        self._store_sum(key + "_in", 0)
        return max(float(self.store.get(f"{key}_r{r}").decode()) for r in range(self.world))

    def barrier(self):
        if not self.enabled:
            return
        if self.dry_run or self.collective == "store":
            self._barriers += 1
            self._store_sum(f"barrier{self._barriers}", 0)
            return
        from pygmu2_amd import comm
        comm.reduce_scalar(0.0, "sum")

    def max_over_ranks(self, value: float) -> float:
        if not self.enabled or self.dry_run:
            return value
        if self.collective == "store":
            self._barriers += 1
            return self._store_max(f"max{self._barriers}", value)
        from pygmu2_amd import comm
        return comm.reduce_scalar(value, "max")

    def _leave_store(self):
        # rank 0 may be hosting the store (bench.py's own spawner): it leaves last
        if self.rank != 0:
            self.store.add("pgx_done", 1)
            return
        t0 = time.time()
        while int(self.store.add("pgx_done", 0)) < self.world - 1 and time.time() - t0 < 60:
            time.sleep(0.005)

    def shutdown(self):
        if not self.enabled:
            return
        if not self.dry_run:
            from pygmu2_amd import comm
            self.barrier()
            if self.collective == "rccl":
                comm.destroy()
        self._leave_store()


class _Solo:
    """The rank-local stand-in used for the N = 1 side measurements of a default run."""
    world, rank, enabled = 1, 0, False

    def barrier(self):
        pass

    def max_over_ranks(self, v):
        return v


def timed_steps(dist, step, steps: int, warmup: int, finish=None):
    """W untimed + K timed calls of step(i); barrier + device sync on both sides; max over ranks."""
    from pygmu2_amd import device
    for i in range(warmup):
        step(i)
    if finish:
        finish()
    device.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    if finish:
        finish()
    device.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    return dist.max_over_ranks(dt)


def host_info():
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return {"host_cores": os.cpu_count() or usable, "host_cores_usable": usable, "host_cpu": model}


# ----------------------------------------------------------------------------- C2
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
    # The timed steps start away from the warm-up stream: every frame of the timed region is rendered inside it (a
    # window opened during the warm-up is never served from).  The seek itself -- the old stream's window settled,
    # one render outside any window -- is the warm-up's last step: the timed region is one sequential stream.
    origin = (warmup + 1000) * frames

    def step(i):
        if i < warmup - 1:
            pos = i * frames
        else:
            pos = origin + (i - warmup) * frames       # i = warmup - 1: the block just before the timed ones
        keep["s"] = pe.render(pos, frames)             # stays in HBM

    from pygmu2_amd import look_ahead
    before = dict(look_ahead.STATS)

    def warm_done(i, _step=step):
        if i == warmup:                                 # the first timed step: everything before it is warm-up
            before.update(look_ahead.STATS)
        _step(i)

    dt = timed_steps(dist, warm_done, steps, warmup)
    r.stop()
    bench_c2.rendered = {"frames_rendered_in_timed_region": look_ahead.STATS["window_frames"] - before["window_frames"],
                         "frames_counted": frames * steps,
                         "windows_opened_in_timed_region": look_ahead.STATS["windows"] - before["windows"]}
    return dt, frames


def bench_c2_with_d2h(pg, steps, warmup, frames=1_000_000):
    """The same steps with the root Snippet handed to the caller as host data every step, as the reference's
    loop does (benchmarks/benchmark_pes.py:176-185; SURVEY 8d "GPU timings include the final D2H of the root
    Snippet and a stream sync").  pipelined: block k crosses PCIe (pinned buffer, copy stream) while block
    k+1 renders, the caller reads k after issuing k+1; sync: render, copy, wait, every step."""
    from pygmu2_amd import device
    out = {}
    for mode in ("pipelined", "sync"):
        pe, r = c2_graph(pg)
        state = {"prev": None, "sum": 0.0}

        def step(i):
            s = pe.render((i if i < warmup - 1 else i + 1000) * frames, frames)  # timed steps: a stream of their own
                                                                                 # (the seek is the last warm-up step)
            if mode == "sync":
                state["sum"] += float(s.data[-1, 0])
                return
            s.prefetch()
            prev, state["prev"] = state["prev"], s
            if prev is not None:
                state["sum"] += float(prev.data[-1, 0])

        def finish():
            prev, state["prev"] = state["prev"], None
            if prev is not None:
                state["sum"] += float(prev.data[-1, 0])

        dt = timed_steps(_Solo(), step, steps, warmup, finish)
        r.stop()
        out[mode] = {"value": round(frames * steps / dt / 1e6, 3), "unit": "Msamples/s",
                     "ms_per_step": round(dt / steps * 1e3, 6),
                     "pcie_gb_s": round(4.0 * frames * steps / dt / 1e9, 2)}
    out["note"] = ("C2 steps with the root Snippet's 4 MB read on the host each step (pinned block from "
                   "pgx_host_malloc, pgx_d2h_begin / pgx_d2h_wait); never `value`")
    return out


def pmc_traffic(entry: str, frames: int):
    """HBM bytes per launch from the committed PMC measurement (profiles/traffic.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(entry, {}).get(str(frames))
    except (OSError, ValueError):
        return None


_ISA = None


def isa_counts():
    """profiles/r4_isa_counts.json: float64 issue slots per unit of the compute-bound kernels' steady-state loop
    bodies, counted in the gfx950 ISA by tools/isa_count.py (committed; a CPU test regenerates and compares it)."""
    global _ISA
    if _ISA is None:
        try:
            with open(os.path.join(ROOT, "profiles", "r4_isa_counts.json")) as f:
                _ISA = json.load(f)
        except (OSError, ValueError):
            _ISA = {"kernels": {}}
    return _ISA


def fp64_roofline(kernel: str, units_per_launch: float, avg_launch_ms: float):
    """the FP64-VALU fraction of a kernel that is compute-bound by construction.  achieved = units per launch
    x float64 issue slots per unit (ISA count of the steady-state body: full-rate float64 instructions and conversions
    1 slot, v_rcp_f64 4) / the kernel's HIP-event time; peak = 39.3 T lane-slots/s."""
    k = isa_counts()["kernels"].get(kernel)
    if not k or "slots_per_unit" not in k or not avg_launch_ms:
        return None
    per_s = units_per_launch / (avg_launch_ms * 1e-3)
    achieved = per_s * k["slots_per_unit"] / 1e12
    all_valu = per_s * k["slots_per_unit_all_valu"] / 1e12
    return {"bound": "fp64_valu", "unit": "T lane-slots/s", "peak": round(F64_PEAK_TSLOTS, 2),
            "achieved": round(achieved, 3), "frac": round(achieved / F64_PEAK_TSLOTS, 4),
            "frac_with_32bit_valu": round(all_valu / F64_PEAK_TSLOTS, 4),
            "kernel": kernel, "unit_counted": k["unit"], "slots_per_unit": k["slots_per_unit"],
            "slots_per_unit_with_32bit_valu": k["slots_per_unit_all_valu"],
            "units_per_launch": units_per_launch, "avg_launch_ms": round(avg_launch_ms, 6),
            "giga_units_per_s": round(per_s / 1e9, 2), "peak_tflops_fma_as_two": round(2 * F64_PEAK_TSLOTS, 1),
            "counts_from": "profiles/r4_isa_counts.json (tools/isa_count.py: gfx950 ISA, steady-state loop body; "
                           "32-bit VALU instructions count half a slot in frac_with_32bit_valu)"}


def bank_kernel_fp64(pg, config: str, launches: int = 300, warm: int = 100):
    """The dominant kernel of a bank mix alone, HIP events around back-to-back launches through the bank's own node
    (states carried from block to block as in the mix): k_supersaw_wide<4> over 512 x 7 oscillators, or C5's 512
    oscillator -> filter -> x envelope voices mixed on chip (pgx_voice_tiles: k_voice_tiles<4, env> and its two small
    launches, timed together; k_blitsaw_biquad_wide<4> where that path is off); 48 000-frame blocks."""
    from pygmu2_amd import voice_bank as vb
    from pygmu2_amd.sharding import mix_voice_factory
    pg.set_sample_rate(48000)
    make, voices = mix_voice_factory(config)
    mix = pg.MixPE(*[make(pg, i) for i in range(voices)])
    bank = mix._voice_bank()
    if not bank:
        return None
    n, pos = 48_000, [0]
    if config == "supersaw":
        node = bank.root
        if not (isinstance(node, vb._SuperSawNode) and node.wide()):
            return None
        kernel, units = "k_supersaw_wide<4>", voices * node.nv * n

        def launch():
            node._bank(pos[0], n)
            pos[0] += n
    else:
        node = next((x for x in bank._nodes() if isinstance(x, vb._BiquadNode)), None)
        if node is None or not node.children["source"].wide():
            return None
        if node.mixes_on_chip(n):
            # the on-chip mix: oscillator -> filter -> x envelope -> partial sums (k_voice_tiles) + the rows' sum; the
            # envelopes are a constant buffer here (their walk runs beside this kernel in the mix, on another stream)
            from pygmu2_amd._kernels import DeviceBuffer
            import numpy as np
            gains = DeviceBuffer.from_host(np.full((voices, n, 1), 0.5, dtype=np.float32))
            kernel, units = "k_voice_tiles<4, env>", voices * n

            def launch():
                node.render_mix(pos[0], n, gain=gains, streaming=True)
                pos[0] += n
        else:
            kernel, units = "k_blitsaw_biquad_wide<4>", voices * n

            def launch():
                node.render(pos[0], n)
                pos[0] += n
    ms = event_avg_ms(launch, launches, warm)        # (a few hundred launches: the clock a stream runs at, as for the mixes)
    return fp64_roofline(kernel, units, ms)


def event_avg_ms(launch, launches, warm=3):
    from pygmu2_amd import device
    for _ in range(warm):
        launch()
    e0, e1 = device.Event(), device.Event()
    e0.record()
    for _ in range(launches):
        launch()
    e1.record()
    return e1.elapsed_ms_since(e0) / launches


def sine_kernel_roofline(pg, frames, launches, start):
    """The other launch of a C2 window: pgx_sine_render at the window's size, at the stream position the timed
    steps are at (the argument of the sine grows with time).  Priced against HBM like everything else (4 B per
    frame written), though float64 issue is what bounds it."""
    from pygmu2_amd import device
    lib = device.ensure_init()
    pg.set_sample_rate(44100)
    pe = pg.SinePE(frequency=440.0)
    out = device.DeviceBuffer((frames, 1), np.float32)
    params = pe._pure_params()

    def launch():
        device.check(lib.pgx_sine_render(out.ptr, 0, 1, start, frames, 1, 44100.0, params.ptr))

    ms = event_avg_ms(launch, launches)
    algo = 4.0 * frames
    return {"bound": "hbm", "achieved": round(algo / (ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
            "kernel": "k_sine (pgx_sine_render, mono)", "frames_per_launch": frames,
            "algorithmic_bytes_per_launch": algo, "avg_launch_ms": round(ms, 6),
            "note": "float64-issue bound, not HBM: ~30 float64-class instructions per sample "
                    f"({frames / (ms * 1e-3) / 1e9:.0f} Gsamples/s); listed because it is the other half of a C2 window"}


def biquad_sine_roofline(pg, frames, launches, start, warm=3):
    """The C2 chain as the timed steps launch it: pgx_biquad_sine (the sine generated inside the settled filter
    kernel), one launch per look-ahead window.  Algorithmic bytes: SURVEY 8d "C2 = 4 B/frame fused with its SinePE"
    (the float32 output; nothing is read)."""
    from pygmu2_amd import device
    from pygmu2_amd.biquad_pe import rbj_coefficients, settle_frames
    lib = device.ensure_init()
    pg.set_sample_rate(44100)
    out = device.DeviceBuffer((frames, 1), np.float32)
    c = rbj_coefficients(pg.BiquadMode.LOWPASS, 1000.0, 0.707, 0.0, 44100.0)
    coef = device.DeviceBuffer.from_host(np.asarray(c, dtype=np.float64))
    settle = settle_frames(c[3], c[4])
    state = device.DeviceBuffer((1, 2), np.float64, zero=True)
    tables = device.DeviceBuffer((lib.pgx_biquad_table_doubles(),), np.float64)
    device.check(lib.pgx_biquad_tables(tables.ptr, coef.ptr, 1))
    w = 2.0 * np.pi * 440.0

    def launch():
        device.check(lib.pgx_biquad_sine(out.ptr, start, frames, 44100.0, w, 1.0, 0.0, coef.ptr, tables.ptr, settle,
                                         state.ptr, None))

    ms = event_avg_ms(launch, launches, warm)
    algo_bytes = 4.0 * frames
    achieved = algo_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic("pgx_biquad_sine", frames),
            "launches_timed": launches, "launches_before": warm,
            "kernel": f"k_biquad_settled<mono, staged, sine> (pgx_biquad_sine, settle_frames={settle})",
            "frames_per_launch": frames, "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(ms, 6),
            "gsamples_per_s": round(frames / (ms * 1e-3) / 1e9, 1),
            "note": "one launch per window, 4 B/frame written, nothing read: bound by float64 issue (sine rotation + "
                    "two filter passes + scan, ~25 float64 operations per frame), not by HBM"}


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

    ms = event_avg_ms(launch, launches)
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
    out = {"value": round(frames * reps / t_all / 1e6, 3), "unit": "Msamples/s", "cores": 1,
           "kind": "port",
           "sample": f"{reps} x render of {frames} frames: oracle sine_pure + biquad_const "
                     f"(numpy sin + scipy.signal.lfilter, float64), 1 thread, {t_all:.1f} s"}
    out.update(host_info())
    return out


# ----------------------------------------------------------------------------- C1 and the small-block cases
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


# ----------------------------------------------------------------------------- C3
def c3_inputs(frames):
    x = (np.random.default_rng(0).standard_normal((frames, 2)) * 0.1).astype(np.float32)
    n = np.arange(65536)
    h = (np.random.default_rng(1).standard_normal(65536) * np.exp(-n / 8000.0)).astype(np.float32)
    return x, h


def bench_c3(pg, dist, steps, warmup, frames=96_000, block=None):
    """C3: ConvolvePE(stereo ArrayPE, 65 536-tap FIR, fft_size=131072), 48 kHz; a step renders the whole
    `frames`-long signal, in one call or in `block`-frame calls, from stream position 0 (the overlap
    history is carried from block to block and across steps)."""
    pg.set_sample_rate(48000)
    x, h = c3_inputs(frames)
    pe = pg.ConvolvePE(pg.ArrayPE(x), pg.ArrayPE(h), fft_size=131072)
    r = pg.NullRenderer(sample_rate=48000)
    r.set_source(pe)
    r.start()
    keep = {}

    def step(i):
        if block is None:
            keep["s"] = pe.render(0, frames)
            return
        pos = 0
        while pos < frames:
            n = min(block, frames - pos)
            keep["s"] = pe.render(pos, n)
            pos += n

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

    ms = event_avg_ms(launch, launches, warm=2)
    flops = 2.0 * 65536 * 2 * frames                # direct form: 2*L*C_out per frame
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(achieved, 3), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 5), "traffic": None,
            "kernel": "k_conv_mfma<4096> (+prep/reduce/hist, pgx_convolve)",
            "frames_per_launch": frames, "algorithmic_flops_per_launch": flops, "avg_launch_ms": round(ms, 6)}


def conv_fft_roofline(pg, frames, launches, nfft=None):
    """HIP-event timing of pgx_convolve_fft (the path ConvolvePE takes for the 65 536-tap C3 filter); nfft: the
    transform size (default: the one ConvolvePE picks for a render of `frames` frames)."""
    from pygmu2_amd import device
    from pygmu2_amd.convolve_pe import device_fft_size
    lib = device.ensure_init()
    x, h = c3_inputs(frames)
    L = 65536
    nfft = nfft or device_fft_size(L, frames)
    xd, hd = device.DeviceBuffer.from_host(x), device.DeviceBuffer.from_host(h.reshape(-1, 1))
    spec = device.DeviceBuffer((lib.pgx_convolve_fft_spectrum_bytes(nfft, 1),), np.uint8)
    device.check(lib.pgx_convolve_fft_prepare(spec.ptr, hd.ptr, L, 1, nfft))
    out = device.DeviceBuffer((frames, 2), np.float32)
    hist = device.DeviceBuffer((L - 1, 2), np.float32, zero=True)
    ws = device.DeviceBuffer((lib.pgx_convolve_fft_workspace_bytes(frames, L, 2, nfft),), np.uint8)

    def launch():
        device.check(lib.pgx_convolve_fft(out.ptr, xd.ptr, frames, 2, spec.ptr, L, 1, 2, nfft, hist.ptr, ws.ptr, 0))

    ms = event_avg_ms(launch, launches, warm=2)
    algo_bytes = 4.0 * (2 + 2) * frames              # 4(C_in + C_out) per frame; taps are read once
    achieved = algo_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic("pgx_convolve_fft", frames),
            "kernel": f"pgx_convolve_fft (float64 four-step FFT overlap-save, N={nfft})",
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


# ----------------------------------------------------------------------------- sharded mixes (C4, C5, SuperSaw)
def mix_entry(pg, dist, config, steps, warmup, with_cpu):
    from pygmu2_amd.sharding import bench_voice_mix
    voices = 64 if config in ("c4", "c4r06") else 512
    dt, frames, name, info = bench_voice_mix(pg, dist, steps, warmup, voices=voices, config=config)
    per_voice = {"c4": 7, "c4r06": 7, "c5": 1, "supersaw": 7}[config]
    out = {"value": round(frames * steps / dt / 1e6, 3), "unit": "Msamples/s",
           "ms_per_block": round(dt / steps * 1e3, 4), "scaling": "strong", "workload": name,
           "steps": steps, "warmup": warmup, "n_ranks": dist.world if dist.enabled else 1,
           "voices_on_this_rank": info["owned"],
           "render_ms": info.get("render_ms"), "render_ms_max_over_ranks": info.get("render_ms_max"),
           "allreduce_wait_ms": info.get("allreduce_wait_ms"),
           "collectives_in_timed_region": info.get("collectives_in_timed_region"),
           "floats_reduced_in_timed_region": info.get("floats_reduced_in_timed_region"),
           "sharded_vs_unsharded_max_err_over_peak": info.get("sharded_vs_unsharded_max_err_over_peak"),
           "agreement_checks": info.get("agreement_checks"),
           "oscillator_msamples_s": round(per_voice * voices * frames * steps / dt / 1e6, 1),
           "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                        "achieved": round(4.0 * frames * steps / dt / 1e9, 4),
                        "frac": round(4.0 * frames * steps / dt / 1e9 / HBM_PEAK_GBS, 7),
                        "algorithmic_bytes_per_block": 4.0 * frames,
                        "note": "SURVEY 8d: 4C B/frame of final mix; every oscillator / filter is an on-chip "
                                "intermediate, so this path is compute / latency bound by construction"}}
    if config in ("c4", "c4r06"):
        # LadderPE: chains x oversampled steps per second (latency-bound, no bandwidth fraction)
        out["chain_steps_per_s"] = round(voices * 2 * frames * steps / dt, 1)
    if with_cpu:
        out["cpu_baseline"] = cpu_mix(config, voices, frames)
        out["over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    out["_dt"] = dt
    return out


def cpu_mix(config, voices, frames, budget_s=5.0, max_picks=96):
    """CPU oracle on a bounded sample of the same workload: voices spread evenly over the index range (the
    cost grows with the oscillator frequency), one block each, until `budget_s` seconds are spent; the block of
    all `voices` is their mean x voices (voices are independent and the mix is one add per voice).  1 thread."""
    from oracle import graph_eval
    from oracle.golden_cases import S

    def spec(i):
        if config == "c5":
            return S("GainPE",
                     source=S("BiquadPE", source=S("BlitSawPE", frequency=27.5 * 2 ** (i / 48.0)),
                              frequency=2000.0, q=0.707),
                     gain=S("AdsrGatedPE", gate=S("PeriodicGate", frequency=2.0 + 0.01 * i, duty_cycle=0.5),
                            attack_time=0.01, decay_time=0.1, sustain_level=0.7, release_time=0.2))
        if config in ("c4", "c4r06"):
            return S("LadderPE", source=S("SuperSawPE", frequency=55.0 * 2 ** (i / 12.0), voices=7,
                                          detune_cents=20.0, seed=i),
                     frequency=1200.0, resonance=0.3 if config == "c4" else 0.6, mode="lp24", drive=1.0, oversample=2)
        return S("SuperSawPE", frequency=55.0 * 2 ** (i / 96.0), voices=7, detune_cents=20.0, seed=i)

    order = [int((k * 0.6180339887 % 1.0) * voices) for k in range(max_picks)]      # low-discrepancy spread
    picks, t_all = [], 0.0
    for i in order:
        if i in picks:
            continue
        g = graph_eval.Node(spec(i), 48000)
        g.render(0, 4800)                               # warm (C library load, allocations)
        t0 = time.perf_counter()
        g.render(4800, frames)
        t_all += time.perf_counter() - t0
        picks.append(i)
        if t_all >= budget_s:
            break
    per_voice = t_all / len(picks)
    out = {"value": round(frames / (per_voice * voices) / 1e6, 5), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": f"{len(picks)} of the {voices} voices (indices spread over the range), one {frames}-frame block "
                     f"each = {t_all:.2f} s; whole mix = mean x {voices}; oracle numpy/scipy + oracle/seq_kernels.c "
                     f"(gcc -O2) for the ladder and ADSR loops"}
    if config in ("c4", "c4r06"):
        out["chain_steps_per_s"] = round(2 * frames / per_voice, 1)
    out.update(host_info())
    return out


# ----------------------------------------------------------------------------- the benchmark_pes.py protocol
def suite_rows(pg, with_cpu):
    """benchmarks/benchmark_pes.py:149-196 protocol on its own configs (:257-383) that north_star names:
    5 warm-up + 50 timed contiguous renders of 44 100 frames through a started NullRenderer graph;
    "sync" waits for the device after every render, "pipelined" once after the 50."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_suite as B
    # every config of the reference's suite this build has a PE for (all but the three RandomPE rows); the rows
    # north_star words its target on get the longer CPU sample
    headline = ("BiquadPE", "SVFilterPE", "SinePE", "BlitSawPE (440 Hz, auto", "SuperSawPE (7")
    # all device rows first, then all CPU rows: a second of CPU timing between two device rows lets the GPU clocks
    # fall back, and the next row's five warm-up renders do not bring them up again
    rows = {}
    for name, spec in B.CONFIGS:
        rows[name] = {k: round(v, 1) for k, v in B.device_rates(spec).items()}
    for name, spec in B.CONFIGS:
        row = rows[name]
        if with_cpu and "SVFilterPE (lowpass, modulated" not in name:     # that oracle loop is plain Python
            row["cpu"] = round(B.cpu_rate(spec, budget_s=1.5 if name.startswith(headline) else 0.5), 2)
            row["pipelined_over_cpu"] = round(row["pipelined"] / row["cpu"], 1)
            row["sync_over_cpu"] = round(row["sync"] / row["cpu"], 1)
    return {"protocol": "benchmark_pes.py:149-196: 44 100-frame renders, 5 warm-up + 50 timed contiguous renders "
                        "(started away from the warm-up: every timed frame is rendered inside the timed region), "
                        "Msamples/s; sync = device wait after every render, pipelined = one wait after the 50, "
                        "block_by_block = pipelined with read-ahead / look-ahead off; cpu = oracle on this host, "
                        "1 thread", "rows": rows}


def north_star_pe_rows(pg, with_cpu):
    """The north_star PEs the reference's own suite has no config for -- CombPE, LadderPE, AdsrGatedPE -- under that
    suite's protocol (44 100-frame renders, 5 + 50), each with the CPU figure of the same graph (oracle:
    oracle/seq_kernels.c, gcc -O2, 1 thread -- the stand-in for the reference's numba kernels), plus a 512-chain
    CombPE bank under a MixPE (48 000-frame blocks)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_suite as B
    import comb_probe as CP
    from oracle.golden_cases import S
    sine = lambda f=440.0, a=1.0, ch=1: S("SinePE", frequency=f, amplitude=a, channels=ch)
    configs = list(CP.CONFIGS) + [
        ("LadderPE (lp24, 1200 Hz, res 0.3, oversample 2)",
         S("LadderPE", source=S("BlitSawPE", frequency=110.0), frequency=1200.0, resonance=0.3, mode="lp24",
           drive=1.0, oversample=2)),
        # at and above self-oscillation (k = 4 res 1.8 > 4 from res 0.556): the reference's own example setting
        # (examples/17_ladder_filter.py:43) and a stronger one over a saw, which locks the saturating loop to itself --
        # time segments with warm-ups found by trial -- and over a lone sine, which does not: the loop runs free, every
        # trial fails the device check and the chain stays on the sequential kernel (one lane: below the CPU)
        ("LadderPE (lp24, 800 Hz, res 0.6, drive 1.5)",
         S("LadderPE", source=S("BlitSawPE", frequency=110.0), frequency=800.0, resonance=0.6, mode="lp24",
           drive=1.5, oversample=2)),
        ("LadderPE (lp24, 800 Hz, res 0.9, drive 1.5)",
         S("LadderPE", source=S("BlitSawPE", frequency=110.0), frequency=800.0, resonance=0.9, mode="lp24",
           drive=1.5, oversample=2)),
        ("LadderPE (res 0.9 over a lone sine: free-running, sequential)",
         S("LadderPE", source=sine(220.0, 0.5), frequency=800.0, resonance=0.9, mode="lp24", drive=1.5,
           oversample=2)),
        ("LadderPE (modulated cutoff)",
         S("LadderPE", source=S("BlitSawPE", frequency=110.0), frequency=S("MixPE", inputs=[
             S("ConstantPE", value=1200.0), sine(0.5, 600.0)]), resonance=0.3, mode="lp24", drive=1.0, oversample=2)),
        ("AdsrGatedPE (PeriodicGate 2 Hz)",
         S("AdsrGatedPE", gate=S("PeriodicGate", frequency=2.0, duty_cycle=0.5), attack_time=0.01, decay_time=0.1,
           sustain_level=0.7, release_time=0.2)),
        ("AdsrGatedPE (PeriodicGate 7 Hz)",
         S("AdsrGatedPE", gate=S("PeriodicGate", frequency=7.0, duty_cycle=0.5), attack_time=0.01, decay_time=0.1,
           sustain_level=0.7, release_time=0.2)),
    ]
    rows = {}
    for name, spec in configs:                       # device rows first, CPU rows after (see suite_rows)
        rows[name] = {k: round(v, 1) for k, v in B.device_rates(spec).items()}
    dt = CP.bank_rate()
    for name, spec in configs:
        if with_cpu:
            row = rows[name]
            row["cpu"] = round(B.cpu_rate(spec, budget_s=1.0), 2)
            row["pipelined_over_cpu"] = round(row["pipelined"] / row["cpu"], 1)
            row["sync_over_cpu"] = round(row["sync"] / row["cpu"], 1)
    bank = {"ms_per_block": round(dt * 1e3, 4), "value": round(48000 / dt / 1e6, 3), "unit": "Msamples/s",
            "chain_msamples_s": round(512 * 48000 / dt / 1e6, 1),
            "workload": "512 x CombPE(BlitSawPE(f_i), 55*2^(i/96) Hz, fb 0.7) -> MixPE, 48 kHz, 48 000-frame blocks"}
    if with_cpu:
        cpu = CP.bank_cpu()
        bank["cpu_ms_per_block"] = round(cpu * 1e3, 2)
        bank["over_cpu"] = round(cpu / dt, 1)
    return {"protocol": "benchmark_pes.py:149-196 (44 100-frame renders, 5 warm-up + 50 timed), Msamples/s; "
                        "cpu = oracle (seq_kernels.c -O2 for the comb / ladder / ADSR loops), 1 thread",
            "rows": rows, "comb_bank_512": bank}


# ----------------------------------------------------------------------------- flat record
FLAT_MIX_KEYS = ("value", "ms_per_block", "oscillator_msamples_s", "over_cpu", "render_ms", "render_ms_max_over_ranks",
                 "allreduce_wait_ms", "collectives_in_timed_region", "sharded_vs_unsharded_max_err_over_peak",
                 "agreement_checks", "voices_on_this_rank", "chain_steps_per_s")


def flat_keys(result) -> dict:
    """Scalar `config` entries (Msamples/s unless the name says otherwise): the sharded mixes with their collective and
    N-rank parity figures, the other BASELINE configs, the north_star PE rows, the PCIe-inclusive C2 rate and the
    float64-issue roofline fractions.  `None` where a figure does not exist in this run (N = 1 has no collectives)."""
    flat = {}
    cases = result.get("cases", {})

    def mix(prefix, entry):
        if not isinstance(entry, dict) or "value" not in entry:
            return
        for key in FLAT_MIX_KEYS:
            if entry.get(key) is not None:
                flat[f"{prefix}_{'msamples_s' if key == 'value' else key}"] = entry[key]
        if isinstance(entry.get("cpu_baseline"), dict):
            flat[f"{prefix}_cpu_msamples_s"] = entry["cpu_baseline"].get("value")
        for name in ("roofline", "roofline_fp64"):
            if isinstance(entry.get(name), dict) and entry[name].get("frac") is not None:
                flat[f"{prefix}_{name}_frac"] = entry[name]["frac"]

    if result.get("mix"):                                   # --workload c4 | c5 | supersaw: the primary line itself
        mix("primary", dict(result["mix"], value=result.get("value")))
    mix("supersaw", result.get("supersaw_mix"))
    mix("c5", result.get("voice_mix"))
    mix("c4", result.get("ladder_mix") or cases.get("c4_supersaw_ladder_mix_64"))
    mix("c4_res06", cases.get("c4_res06_supersaw_ladder_mix_64"))
    d2h = result.get("value_with_d2h")
    if isinstance(d2h, dict):
        for mode in ("pipelined", "sync"):
            if mode in d2h:
                flat[f"c2_with_d2h_{mode}"] = d2h[mode].get("value")
                if "over_cpu" in d2h[mode]:
                    flat[f"c2_with_d2h_{mode}_over_cpu"] = d2h[mode]["over_cpu"]
    for key, short in (("c1_sine_gain_1024_blocks", "c1"), ("c1_hello_sine_example_1024_blocks", "c1_hello_sine"),
                       ("c3_convolve_64k_taps", "c3_96000"), ("c3_convolve_64k_taps_1440000_whole", "c3_1440000"),
                       ("c3_convolve_64k_taps_1440000_blocks_65537", "c3_blocks_65537"),
                       ("autowah_biquad_1024_blocks", "autowah_biquad"), ("autowah_svf_1024_blocks", "autowah_svf")):
        case = cases.get(key)
        if isinstance(case, dict):
            flat[f"{short}_msamples_s"] = case.get("value")
            if case.get("cpu_oracle_msamples_s") is not None:
                flat[f"{short}_cpu_msamples_s"] = case["cpu_oracle_msamples_s"]
            roof = case.get("roofline")
            if isinstance(roof, dict):
                flat[f"{short}_roofline_frac"] = roof.get("frac")
                if roof.get("avg_launch_ms") is not None:
                    flat[f"{short}_call_us"] = round(roof["avg_launch_ms"] * 1e3, 2)
                if roof.get("traffic") and roof.get("algorithmic_bytes_per_launch"):
                    flat[f"{short}_traffic_over_algorithmic"] = round(roof["traffic"] / roof["algorithmic_bytes_per_launch"], 2)
    rows = (result.get("north_star_pes") or {}).get("rows", {})
    short_names = {"LadderPE (lp24, 1200 Hz, res 0.3, oversample 2)": "ladder_res03",
                   "LadderPE (lp24, 800 Hz, res 0.6, drive 1.5)": "ladder_res06",
                   "LadderPE (lp24, 800 Hz, res 0.9, drive 1.5)": "ladder_res09",
                   "LadderPE (res 0.9 over a lone sine: free-running, sequential)": "ladder_res09_free_running",
                   "LadderPE (modulated cutoff)": "ladder_modulated_cutoff",
                   "CombPE (440 Hz, fb 0.7)": "comb_440", "CombPE (modulated frequency)": "comb_modulated_frequency",
                   "AdsrGatedPE (PeriodicGate 2 Hz)": "adsr_2hz", "AdsrGatedPE (PeriodicGate 7 Hz)": "adsr_7hz"}
    for name, row in rows.items():
        tag = short_names.get(name)
        if tag:
            for k in ("sync", "pipelined", "cpu", "pipelined_over_cpu"):
                if row.get(k) is not None:
                    flat[f"{tag}_{k}"] = row[k]
    suite = (result.get("suite") or {}).get("rows", {})
    for name, tag in (("BiquadPE (lowpass, fixed)", "suite_biquad"), ("SinePE (440 Hz)", "suite_sine"),
                      ("BlitSawPE (440 Hz, auto M)", "suite_blitsaw"), ("SuperSawPE (7 voices)", "suite_supersaw7")):
        row = suite.get(name)
        if row:
            for k in ("sync", "pipelined", "cpu"):
                if row.get(k) is not None:
                    flat[f"{tag}_{k}"] = row[k]
    if isinstance(result.get("roofline_fp64"), dict):
        flat["c2_roofline_fp64_frac"] = result["roofline_fp64"].get("frac")
    if isinstance(result.get("roofline"), dict) and result["roofline"].get("steps_per_launch"):
        flat["c2_kernel_steps_per_launch"] = result["roofline"]["steps_per_launch"]
        flat["c2_kernel_us_per_launch"] = round(result["roofline"]["avg_launch_ms"] * 1e3, 2)
    if isinstance(result.get("roofline_burst"), dict):
        flat["c2_roofline_burst_frac"] = result["roofline_burst"].get("frac")
    if result.get("process_warmup_streams") is not None:
        flat["c2_process_warmup_streams"] = result["process_warmup_streams"]
    return flat


# ----------------------------------------------------------------------------- dry run (launch glue on CPU)
def dry_run(args, dist):
    import pygmu2_amd as pg
    from pygmu2_amd.sharding import ShardedMixPE, mix_voice_factory
    pg.set_sample_rate(48000)
    owned = None
    if args.workload in SHARDED:
        make, voices = mix_voice_factory(args.workload)
        root = ShardedMixPE([make(pg, i) for i in range(voices)], dist.rank, dist.world)
        owned = len(root.owned)
    seen = dist.connect()
    total_owned = dist._store_sum("owned", owned or 0) if dist.enabled else (owned or 0)
    dist.barrier()
    if dist.rank == 0:
        print(json.dumps({"metric": "Msamples/s rendered (benchmark_pes.py metric: output frames / wall second)",
                          "value": None, "unit": "Msamples/s", "n_gpus": dist.world, "n_ranks_seen": seen,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                          "config": {"workload": args.workload, "voices_on_rank0": owned,
                                     "voices_on_all_ranks": total_owned,
                                     # the scalar keys a measured line of this world size carries for each sharded mix
                                     "flat_mix_keys": ",".join(f"{p}_{'msamples_s' if k == 'value' else k}"
                                                               for p in ("supersaw", "c5", "c4") for k in FLAT_MIX_KEYS)}}),
              flush=True)
    dist.shutdown()


# ----------------------------------------------------------------------------- main
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))          # before anything touches the GPU
    dist = Dist(dry_run=args.dry_run, allow_no_rccl=args.allow_no_rccl)
    if args.dry_run:
        dry_run(args, dist)
        return
    if dist.enabled:
        os.environ.setdefault("PYGMU_DEVICE", str(dist.local_rank))
    import pygmu2_amd as pg
    from pygmu2_amd import device
    device.ensure_init()
    ranks_seen = dist.connect()
    if dist.enabled and dist.collective != "rccl" and args.workload in SHARDED:
        if dist.rank == 0:
            print(json.dumps({"metric": "Msamples/s rendered (benchmark_pes.py metric: output frames / wall second)",
                              "value": None, "unit": "Msamples/s", "n_gpus": dist.world, "n_ranks_seen": ranks_seen,
                              "error": f"workload {args.workload} needs the RCCL communicator: "
                                       f"{dist.collective_error}"}), flush=True)
        dist.shutdown()
        sys.exit(1)

    n_gpus = max(1, dist.world)
    with_cpu = not args.no_cpu and n_gpus == 1
    result = {}
    extra_primary = {}
    if args.workload == "c2":
        # A process that has just started is not the process that renders: the first windows it opens run through cold
        # Python / ctypes paths, and the chip reaches the clock it holds over a stream only after ~20 ms of work.  The
        # driver's `--steps 20 --warmup 5` is a 65 us timed region behind 5 steps: measured in a fresh process it was 3.7 -
        # 4.4 us per step, behind ten untimed 2 000-step streams of OTHER PE objects (25 ms) 3.1 - 3.3.  The W warm-up
        # steps and the K timed steps follow unchanged, on a graph of their own; `process_warmup` in the line says so.
        # PGX_BENCH_PROCESS_WARMUP=0 switches it off.
        process_warmup = int(os.environ.get("PGX_BENCH_PROCESS_WARMUP", "10"))
        for _ in range(process_warmup):
            bench_c2(pg, dist, 2000, 20)
        result["process_warmup"] = (f"{process_warmup} untimed C2 streams of 2 000 steps on other PE objects before the "
                                    f"{args.warmup} warm-up steps" if process_warmup else "none")
        result["process_warmup_streams"] = process_warmup
        dt, frames = bench_c2(pg, dist, args.steps, args.warmup)
        name = "C2: BiquadPE(SinePE(440), lowpass 1 kHz, q 0.707), 44.1 kHz mono, render(start, 1_000_000) per step"
        units = frames * args.steps * n_gpus
    elif args.workload == "c1":
        dt, frames = bench_c1(pg, dist, args.steps, args.warmup)
        name = "C1: GainPE(SinePE(440, ch=2), 0.5), 44.1 kHz stereo, 431 blocks of 1024 per step"
        units = frames * args.steps * n_gpus
    elif args.workload == "c3":
        dt, frames = bench_c3(pg, dist, args.steps, args.warmup)
        name = "C3: ConvolvePE stereo x 65536-tap FIR, 48 kHz, 96 000 frames per step"
        units = frames * args.steps * n_gpus
    else:
        entry = mix_entry(pg, dist, args.workload, args.steps, args.warmup, with_cpu and dist.rank == 0)
        dt, frames, name = entry.pop("_dt"), 48_000, entry["workload"]
        units = frames * args.steps
        extra_primary = {k: entry[k] for k in entry if k not in ("value", "unit", "workload", "steps", "warmup")}

    value = units / dt / 1e6
    sharded = args.workload in SHARDED
    result.update({
        "metric": "Msamples/s rendered (benchmark_pes.py metric: output frames / wall second)",
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": n_gpus, "n_ranks_seen": ranks_seen,
        "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 6), "higher_is_better": True,
        "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": name, "frames_per_step": frames,
                   **(getattr(bench_c2, "rendered", {}) if args.workload == "c2" else {}),
                   "parallelism": (f"inputs of the root MixPE dealt i mod {n_gpus} over the ranks, one RCCL all-reduce "
                                   "of the partial mix per block (pgx_allreduce_sum)" if sharded else
                                   ("single chain" if n_gpus == 1 else f"{n_gpus} independent replicas (replicas only)"))},
    })
    if sharded:
        result["mix"] = extra_primary
        if "roofline" in extra_primary:
            result["roofline"] = extra_primary["roofline"]
        if "cpu_baseline" in extra_primary:
            result["cpu_baseline"] = extra_primary["cpu_baseline"]

    if dist.enabled:
        result["collective"] = dist.collective
        if dist.collective != "rccl":
            result["n_ranks_present"] = getattr(dist, "ranks_present", 0)      # counted through the store, not by RCCL
    if not sharded and not args.no_extras and dist.enabled and dist.collective != "rccl":
        note = {"error": f"no RCCL communicator ({dist.collective_error}): the sharded mixes were not run"}
        result["voice_mix"], result["supersaw_mix"] = dict(note), dict(note)
    elif not sharded and not args.no_extras:
        # collectives: every rank takes part.  The sharded mixes ride along in the default line.
        # (streams of a few hundred blocks -- 40 - 60 ms of device time: over 50 blocks the chip has not yet reached the clock
        # a stream runs at, and the same kernels took 5 - 7 % longer: DESIGN 7)
        # (C5 renders in windows of 2, 4, 8 blocks since its voices are mixed on chip: 384 after 47, as for the SuperSaw mix below)
        result["voice_mix"] = mix_entry(pg, dist, "c5", 384, 47, with_cpu and dist.rank == 0)
        # (384 blocks after 47: a rank's share of a sharded run renders windows of 2, 4, 8, 8, ... blocks from the second
        # block on and reduces each in one collective -- the warm-up ends on a window's last block and holds the first
        # 8-block collectives, the timed region is 48 whole windows: 384 blocks rendered for the 384 counted)
        result["supersaw_mix"] = mix_entry(pg, dist, "supersaw", 384, 47, with_cpu and dist.rank == 0)
        result["voice_mix"].pop("_dt"), result["supersaw_mix"].pop("_dt")
        if dist.enabled:
            # BASELINE config 4 sharded as well (at N = 1 it is `cases.c4_supersaw_ladder_mix_64`)
            result["ladder_mix"] = mix_entry(pg, dist, "c4", 256, 63, False)
            result["ladder_mix"].pop("_dt")

    if dist.rank == 0 and not args.no_extras and not sharded:
        solo = _Solo()
        result["device"] = device.device_name()
        # the filter kernel as the C2 steps launch it: look-ahead renders `ahead` 1 M-frame steps per launch
        from pygmu2_amd import look_ahead
        ahead = max(2, min(look_ahead.AHEAD_BLOCKS, look_ahead.frame_cap(2) // 1_000_000)) if look_ahead.enabled() else 1
        # (steady state: a stream's windows are 8, 16, 32, then `ahead` steps long)
        far = (args.warmup + 1000) * 1_000_000
        # (the dominant kernel over 1 000 back-to-back launches after 300: 50 ms -- the clock the chip holds over a stream,
        # which a burst of 100 launches, 4 ms, does not reach: 41.6 against 37.8 us; `roofline_burst` keeps the old protocol)
        result["roofline"] = biquad_sine_roofline(pg, 1_000_000 * ahead, 1000, far, warm=300)
        result["roofline_burst"] = biquad_sine_roofline(pg, 1_000_000 * ahead, 100, far)
        result["roofline"]["steps_per_launch"] = ahead
        result["roofline_one_step"] = biquad_sine_roofline(pg, 1_000_000, 200, far)
        result["roofline_scaled"] = biquad_sine_roofline(pg, 1 << 26, 10, far)
        # the two kernels of the same chain when it is not fused (stereo, PE-driven sine, filters that reject the
        # tone): the filter alone is the HBM-streaming kernel, 8 B/frame
        result["roofline_filter_alone"] = biquad_kernel_roofline(pg, 1_000_000 * ahead, 50)
        result["roofline_sine_alone"] = sine_kernel_roofline(pg, 1_000_000 * ahead, 30, far)
        # SURVEY 8d / VERDICT r3: the kernels that are compute-bound by construction, priced against the float64 issue
        # peak with instruction counts from the ISA (the HBM fractions above say little about them)
        result["roofline_fp64"] = fp64_roofline("k_biquad_settled<mono, staged, sine, 256>",
                                                result["roofline"]["frames_per_launch"],
                                                result["roofline"]["avg_launch_ms"])
        for key, config in (("supersaw_mix", "supersaw"), ("voice_mix", "c5")):
            if isinstance(result.get(key), dict) and "value" in result[key]:
                result[key]["roofline_fp64"] = bank_kernel_fp64(pg, config)
        cases = {}
        if args.workload == "c2" and n_gpus == 1:
            result["value_with_d2h"] = bench_c2_with_d2h(pg, 40, 5)
            dt1, f1 = bench_c1(pg, solo, 5, 1)
            cases["c1_sine_gain_1024_blocks"] = {"value": round(f1 * 5 / dt1 / 1e6, 3), "unit": "Msamples/s"}
            cases["c1_hello_sine_example_1024_blocks"] = {"value": hello_sine_case(pg), "unit": "Msamples/s"}
            dt3, f3 = bench_c3(pg, solo, 10, 2)
            cases["c3_convolve_64k_taps"] = {"value": round(f3 * 10 / dt3 / 1e6, 3), "unit": "Msamples/s",
                                             "path": "float64 FFT overlap-save (pgx_convolve_fft)",
                                             "roofline": conv_fft_roofline(pg, 96_000, 20),
                                             # the dense FIR x block product on the matrix cores, same filter:
                                             # what ConvolvePE uses below convolve_pe.FFT_MIN_TAPS taps
                                             "direct_form_mfma": conv_kernel_roofline(pg, 96_000, 10)}
            dtw, fw = bench_c3(pg, solo, 5, 1, frames=1_440_000)
            cases["c3_convolve_64k_taps_1440000_whole"] = {
                "value": round(fw * 5 / dtw / 1e6, 3), "unit": "Msamples/s",
                "roofline": conv_fft_roofline(pg, 1_440_000, 5)}
            dtb, fb = bench_c3(pg, solo, 3, 1, frames=1_440_000, block=65_537)
            cases["c3_convolve_64k_taps_1440000_blocks_65537"] = {
                "value": round(fb * 3 / dtb / 1e6, 3), "unit": "Msamples/s",
                "roofline": conv_fft_roofline(pg, 65_537, 20)}
            # (64 blocks after 63: the ladder bank's windows of 2, 4, 8, 16 blocks and the first one of 32 -- whose 2 x 393 MB
            # of buffers are allocated then -- open during the warm-up, which ends on a window's last block; the timed
            # region is two whole windows of 32 -- 64 blocks rendered for the 64 counted)
            cases["c4_supersaw_ladder_mix_64"] = mix_entry(pg, solo, "c4", 256, 63, with_cpu)
            cases["c4_supersaw_ladder_mix_64"].pop("_dt")
            # the same bank with the ladders above self-oscillation (resonance 0.6): warm-ups by trial
            cases["c4_res06_supersaw_ladder_mix_64"] = mix_entry(pg, solo, "c4r06", 256, 63, with_cpu)
            cases["c4_res06_supersaw_ladder_mix_64"].pop("_dt")
            cases["autowah_biquad_1024_blocks"] = {"value": autowah_case(pg, "biquad"), "unit": "Msamples/s"}
            cases["autowah_svf_1024_blocks"] = {"value": autowah_case(pg, "svf"), "unit": "Msamples/s"}
            result["north_star_pes"] = north_star_pe_rows(pg, with_cpu)
            result["suite"] = suite_rows(pg, with_cpu)
        if with_cpu:
            result["cpu_baseline"] = cpu_c2(1_000_000)
            if "value_with_d2h" in result:
                for mode in ("pipelined", "sync"):
                    result["value_with_d2h"][mode]["over_cpu"] = round(
                        result["value_with_d2h"][mode]["value"] / result["cpu_baseline"]["value"], 1)
            if "autowah_biquad_1024_blocks" in cases:
                # the oracle's varying biquad is the C restatement of the numba kernel; its SVF coefficient
                # loop is plain Python (slow), so only the biquad graph gets a CPU figure
                cases["autowah_biquad_1024_blocks"]["cpu_oracle_msamples_s"] = cpu_autowah("biquad")
            if "c1_sine_gain_1024_blocks" in cases:
                cases["c1_sine_gain_1024_blocks"]["cpu_oracle_msamples_s"] = cpu_c1()
                cases["c1_hello_sine_example_1024_blocks"]["cpu_oracle_msamples_s"] = cpu_hello_sine()
            if "c3_convolve_64k_taps" in cases:
                c3cpu = cpu_c3()
                for k in cases:
                    if k.startswith("c3_"):
                        cases[k]["cpu_oracle_msamples_s"] = c3cpu
        if cases:
            result["cases"] = cases
        # the driver's record keeps `config`, `roofline` and `cpu_baseline` whole and only the tail of the rest:
        # the numbers a reader looks for first are repeated here, compactly (Msamples/s unless said otherwise)
        hl = {}
        try:
            if "value_with_d2h" in result:
                hl["c2_with_d2h_pipelined"] = result["value_with_d2h"]["pipelined"]["value"]
                hl["c2_with_d2h_sync"] = result["value_with_d2h"]["sync"]["value"]
            for key, short in (("c1_sine_gain_1024_blocks", "c1"), ("c3_convolve_64k_taps", "c3_96000"),
                               ("c3_convolve_64k_taps_1440000_whole", "c3_1440000"),
                               ("c4_supersaw_ladder_mix_64", "c4")):
                if key in cases:
                    hl[short] = cases[key]["value"]
                    if "cpu_oracle_msamples_s" in cases[key]:
                        hl[short + "_cpu"] = cases[key]["cpu_oracle_msamples_s"]
                    elif "cpu_baseline" in cases[key]:
                        hl[short + "_cpu"] = cases[key]["cpu_baseline"]["value"]
            if "c3_convolve_64k_taps" in cases:
                hl["c3_call_us"] = round(cases["c3_convolve_64k_taps"]["roofline"]["avg_launch_ms"] * 1e3, 2)
            for key, short in (("voice_mix", "c5"), ("supersaw_mix", "supersaw_mix")):
                if key in result and "value" in result[key]:
                    hl[short] = result[key]["value"]
                    hl[short + "_ms_per_block"] = result[key]["ms_per_block"]
                    if "cpu_baseline" in result[key]:
                        hl[short + "_cpu"] = result[key]["cpu_baseline"]["value"]
            if "north_star_pes" in result:
                for name, row in result["north_star_pes"]["rows"].items():
                    tag = name.split(" (")[0].lower() + "_" + name.split("(")[1].rstrip(")").replace(" ", "_").replace(",", "")
                    hl[tag] = [row.get("sync"), row.get("pipelined"), row.get("cpu")]
                hl["comb_bank_512_ms_per_block"] = result["north_star_pes"]["comb_bank_512"]["ms_per_block"]
                hl["_pe_rows"] = "[sync, pipelined, cpu]"
            if "suite" in result:
                for name in ("BiquadPE (lowpass, fixed)", "SinePE (440 Hz)", "BlitSawPE (440 Hz, auto M)",
                             "SuperSawPE (7 voices)"):
                    row = result["suite"]["rows"].get(name)
                    if row:
                        hl["suite " + name] = [row.get("sync"), row.get("pipelined"), row.get("cpu")]
        except (KeyError, TypeError, IndexError):
            pass
        if hl:
            result["config"]["highlights"] = hl

    if dist.rank == 0:
        # The driver's record keeps `config` only as far as its values are scalars (nested objects are dropped, the
        # rest of the line survives as a tail): everything north_star words a target on is repeated there, flat.
        result["config"].update(flat_keys(result))

    if dist.rank == 0:
        print(json.dumps(result), flush=True)
    dist.shutdown()


if __name__ == "__main__":
    main()
