"""
Multi-GPU rendering: shard the inputs of a MixPE over ranks, reduce the partial mixes.

Voices / PE sub-graphs feeding a MixPE share nothing (a stateful PE may have only one
sink, renderer.py:379-384) and MixPE is a plain sum (mix_pe.py:91-94), so the K inputs
are dealt round-robin over the G ranks (input i -> rank i mod G), every rank renders and
mixes its own shard entirely in its own HBM, and ONE all-reduce(sum) per rendered block
combines the partial mixes: RCCL over xGMI through the library's own entry points
(pgx_allreduce_sum, include/pygmu_hip.h; communicator set up by pygmu2_amd.comm), or gloo on host
payloads for the CPU tests of the multi-rank logic.  One process per GPU; any launcher that
sets RANK / LOCAL_RANK / WORLD_SIZE will do (torchrun, bench.py's own spawner).

The all-reduce adds the partial mixes in an order that differs from the reference's
left-to-right float32 sum, so sharded output matches the unsharded one to ~1e-7 of peak
(inside the 1e-5 parity budget), not bit for bit.
"""

from __future__ import annotations

import os

import numpy as np

from .extent import Extent
from .mix_pe import MixPE
from .processing_element import ProcessingElement
from .snippet import Snippet


WINDOW_COLLECTIVES = os.environ.get("PGX_WINDOW_COLLECTIVES", "1") != "0"   # a bank window = one collective


def shard_indices(n_inputs: int, rank: int, world: int) -> list[int]:
    """Indices of the MixPE inputs owned by `rank` (round-robin)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    return list(range(rank, n_inputs, world))


class _Silence(ProcessingElement):
    """Stand-in for a rank that owns no input: contributes zeros to the reduction."""

    def __init__(self, channels: int):
        self._channels = channels

    def inputs(self):
        return []

    def is_pure(self):
        return True

    def channel_count(self):
        return self._channels

    def _compute_extent(self):
        return Extent(0, 0)

    def _render(self, start, duration):
        return Snippet.from_zeros(start, duration, self._channels)


class RcclReducer:
    """all-reduce(sum) of a Snippet payload on RCCL through the C ABI (pgx_allreduce_sum): the product path.

    Out of place into a fresh buffer: the local payload may be storage somebody else keeps (a CachePE memo,
    a row view of a read-ahead window, a pass-through Snippet), so it is only read.  Every rank issues the
    same collective whatever its local payload was (a rank that owns nothing uploads its zeros).  The host
    does not block: the collective is ordered behind the library stream, and the returned Snippet carries a
    `ready` hook that orders the library stream behind the collective only when the payload is USED --
    rendering the next block overlaps this block's all-reduce over xGMI.  A result dropped unread makes nobody
    wait: its buffers (and the local payload) are parked and released by a later wait (all_reduce).
    """

    def __init__(self):
        from . import comm, device
        if not comm.initialised():
            raise RuntimeError("no RCCL communicator: call pygmu2_amd.comm.init(...) on every rank first")
        self._device = device
        self._lib = device.ensure_init()
        self.rank, self.world = comm.info()
        self.calls = 0                # collectives issued, floats reduced (bench.py reports them)
        self.floats = 0
        self._parked = []             # (ticket, buffers kept alive) of results dropped unread, oldest first
        self._comm = comm

    PARK_MAX = 8                      # results dropped unread before the library stream is made to wait for them

    def fold(self, *words) -> None:
        """Facts every rank shares about the collectives that follow: part of the sequence hash the library's
        agreement checks compare across the ranks (a rank out of step fails instead of hanging the communicator)."""
        self._comm.fold_check(*words)

    def checks(self) -> int:
        return self._comm.stats()["checks"]

    def all_reduce(self, snippet: Snippet) -> Snippet:
        import ctypes as C
        dev, lib = self._device, self._lib
        src = snippet.dev
        self.calls += 1
        self.floats += src.nbytes // 4
        out = dev.DeviceBuffer(src.shape, np.float32)
        ticket = C.c_int64(0)
        dev.check(lib.pgx_allreduce_sum(out.ptr, src.ptr, src.nbytes // 4, C.byref(ticket)), "pgx_allreduce_sum")
        hold = [snippet, out]
        parked = self._parked

        def ready():
            # the payload is about to be used: the library stream waits for this collective -- and with it for every
            # earlier one (one in-order collective stream), so whatever was parked before it may go back to the pool
            dev.check(lib.pgx_allreduce_wait(ticket.value), "pgx_allreduce_wait")
            hold.clear()
            while parked and parked[0][0] <= ticket.value:
                parked.pop(0)[1].clear()

        def dropped():
            # nobody looked at this block (a pipelined caller that keeps only the latest): nothing has to wait for it
            # NOW -- a wait is a barrier packet on the library stream, and a stalled queue costs ~17 us to wake -- but its
            # buffers must outlive the collective: they are parked, and one wait per PARK_MAX blocks releases them
            parked.append((ticket.value, hold))
            if len(parked) >= self.PARK_MAX:
                newest = parked[-1][0]
                dev.check(lib.pgx_allreduce_wait(newest), "pgx_allreduce_wait")
                for _, h in parked:
                    h.clear()
                parked.clear()
        ready.on_drop = dropped
        return Snippet(snippet.start, out, ready=ready)

    def __del__(self):
        try:
            if self._parked:
                self._device.check(self._lib.pgx_allreduce_wait(self._parked[-1][0]), "pgx_allreduce_wait")
                self._parked.clear()
        except Exception:
            pass


class TorchReducer:
    """all-reduce(sum) through an initialised torch.distributed group: gloo on host payloads (the CPU tests
    of the multi-rank logic), or -- PYGMU_REDUCER=torch -- torch's own RCCL binding on device payloads.
    The path is chosen from the group's backend, never from where this rank's payload happens to live, so
    that every rank issues the same collective."""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.on_device = "nccl" in str(dist.get_backend())
        self._lib_stream = None
        self._comm_stream = None
        self._hash = 0                # sequence hash: every count and every folded fact so far
        self._tickets = 0
        self._checks = 0
        self.calls = 0                # collectives issued, floats reduced (as RcclReducer: bench.py reports them)
        self.floats = 0

    CHECK_FIRST, CHECK_EVERY = 8, 16  # as pgx_comm.hip: the first tickets and every 16th after them are checked

    def fold(self, *words) -> None:
        for w in words:
            self._hash = hash((self._hash, int(w))) & ((1 << 62) - 1)      # ints and tuples of ints hash alike everywhere

    def checks(self) -> int:
        return self._checks

    def _agree(self, n: int) -> None:
        """The same agreement check pgx_comm.hip makes in front of a collective: all ranks reduce {n, -n, h, -h} with
        max; they agree exactly when max(n) == -max(-n) and max(h) == -max(-h).  A fixed-size collective cannot
        mismatch, so a rank out of step raises on every rank instead of hanging the group."""
        self.fold(n)
        self._tickets += 1
        if not (self._tickets <= self.CHECK_FIRST or self._tickets % self.CHECK_EVERY == 0):
            return
        torch, dist = self.torch, self.dist
        t = torch.tensor([n, -n, self._hash, -self._hash], dtype=torch.int64)
        if self.on_device:
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        v = [int(x) for x in t.cpu()]
        self._checks += 1
        if v[0] != -v[1] or v[2] != -v[3]:
            raise RuntimeError(
                f"ranks out of step at collective {self._tickets}: this rank reduces {n} floats, the ranks' counts span "
                f"{-v[1]} .. {v[0]}" + (" (same count, different history of pulls)" if v[0] == -v[1] else "")
                + " -- every rank must make the same sequence of pulls")

    def _streams(self):
        if self._lib_stream is None:
            from . import device as _dev
            handle = _dev.ensure_init().pgx_stream_handle()
            self._lib_stream = self.torch.cuda.ExternalStream(int(handle))
            self._comm_stream = self.torch.cuda.Stream()
        return self._lib_stream, self._comm_stream

    def all_reduce(self, snippet: Snippet) -> Snippet:
        torch, dist = self.torch, self.dist
        self.calls += 1
        self.floats += int(snippet.duration) * int(snippet.channels)
        self._agree(int(snippet.duration) * int(snippet.channels))
        if self.on_device:
            from . import device as _dev
            src = snippet.dev                           # a host payload (zeros of an idle rank) is uploaded
            buf = _dev.DeviceBuffer(src.shape, np.float32)
            _dev.check(_dev.ensure_init().pgx_memcpy_d2d(buf.ptr, src.ptr, src.nbytes), "pgx_memcpy_d2d")
            lib_stream, comm = self._streams()
            t = torch.as_tensor(buf, device="cuda")     # zero-copy via __cuda_array_interface__
            comm.wait_stream(lib_stream)                # the collective starts after the local mix is complete
            with torch.cuda.stream(comm):
                work = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

            def ready():
                with torch.cuda.stream(lib_stream):
                    work.wait()                         # stream-level wait: the host is not blocked
            return Snippet(snippet.start, buf, ready=ready)
        t = torch.from_numpy(np.ascontiguousarray(snippet.data).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return Snippet(snippet.start, t.numpy())


def default_reducer():
    """The C-ABI RCCL communicator when one exists (pygmu2_amd.comm.init); otherwise an initialised
    torch.distributed group (gloo: CPU tests)."""
    import os
    from . import comm
    if os.environ.get("PYGMU_REDUCER", "") != "torch":
        try:
            if comm.initialised():
                return RcclReducer()
        except RuntimeError:                            # library not built: only the host-side tests get here
            pass
    return TorchReducer()


class ShardedMixPE(ProcessingElement):
    """
    MixPE whose inputs are split over `world` ranks.

    Every rank constructs it with the FULL input list (cheap: PE objects are host-side
    descriptions; device state is created lazily, on first render, only for the inputs a rank
    owns) and renders the same (start, duration) windows; the result on every rank is the
    full mix.  `local_mixer(inputs)` builds the rank-local mix PE (default: MixPE, which in
    turn batches identical voices into a voice bank).
    """

    def __init__(self, inputs, rank: int, world: int, reducer=None, local_mixer=None):
        inputs = list(inputs)
        if len(inputs) < 2:
            raise ValueError("ShardedMixPE requires at least 2 inputs")
        self._all_inputs = inputs
        self._rank, self._world = rank, world
        self._owned = [inputs[i] for i in shard_indices(len(inputs), rank, world)]
        self._channels = inputs[0].channel_count()
        mixer = local_mixer or (lambda pes: MixPE(*pes))
        if len(self._owned) >= 2:
            self._local = mixer(self._owned)
        elif len(self._owned) == 1:
            self._local = self._owned[0]
        else:
            self._local = _Silence(self._channels or 1)
        self._reducer = reducer
        self._folded_config = False
        self._windows = None          # decided once, from what every rank knows (see _whole_windows)
        self._reduced = None          # (local window buffer, its reduced Snippet, that Snippet's buffer)
        if world > 1 and isinstance(self._local, MixPE) and self._whole_windows():
            self._local.__dict__["_mix_windows"] = True       # the bank makes its windows at the level of the mix

    owned = property(lambda self: self._owned)
    local = property(lambda self: self._local)

    def inputs(self):
        return [self._local]

    def is_pure(self):
        return False          # collective: every rank must issue the same render sequence

    def channel_count(self):
        return self._channels

    def _compute_extent(self):
        ext = self._all_inputs[0].extent()
        for pe in self._all_inputs[1:]:
            ext = ext.union(pe.extent())
        return ext

    def _whole_windows(self) -> bool:
        """May a window of the rank-local bank be reduced in one collective?  Every rank has to answer alike -- a
        rank that reduces a window while another reduces a block hangs the communicator -- so the answer comes
        from what all ranks know: the full input list, the world size and voice_bank's window rule (banks of
        MIN_VOICES .. BANK_WINDOW_MAX_VOICES SuperSaw instances, or ladders over them, open windows of 2, 4, 8 blocks
        from the second block of a stream of equal blocks on; which blocks those are depends on the sequence of pulls
        alone, and that is the same on every rank by ShardedMixPE's contract)."""
        if self._windows is None:
            from . import voice_bank as vb
            n, world = len(self._all_inputs), self._world
            sig = vb._signature(self._all_inputs[0])
            seen = set()
            self._windows = bool(
                WINDOW_COLLECTIVES and vb.BANK_WINDOWS and not vb.BANK_WINDOWS_ANY_ROOT
                and n // world >= vb.MIN_VOICES and -(-n // world) <= vb.BANK_WINDOW_MAX_VOICES
                and -(-n // world) < vb.FUSED_SUPERSAW_MIN
                and sig is not None and (sig[0] == "supersaw" or (sig[0] == "ladder" and sig[1][0] == "supersaw"))
                and all(vb._signature(pe) == sig and vb._collect_ids(pe, seen) for pe in self._all_inputs))
            # Round 4: banks whose voices are mixed on chip (C5: GainPE(BiquadPE(BlitSawPE), gain=<envelope>)) run in
            # windows too (voice_bank.VOICE_TILE_WINDOWS).  Whether a bank takes that path is a matter of its voices'
            # parameters -- every oscillator wide-capable with a closed-form carry, every filter settling inside a tile --
            # and of its size; a rank answers for ALL inputs (voice_bank.on_chip_mix_rule, host-side: if every voice qualifies,
            # every rank's share does) and for the smallest share.
            if (not self._windows and WINDOW_COLLECTIVES and vb.BANK_WINDOWS and not vb.BANK_WINDOWS_ANY_ROOT and vb.VOICE_TILES
                    and vb.VOICE_TILE_WINDOWS and n // world >= max(vb.MIN_VOICES, vb.VOICE_TILES_MIN_VOICES)
                    and sig is not None and sig[0] in ("gain_pe", "biquad")):
                seen = set()
                self._windows = bool(all(vb._signature(pe) == sig and vb._collect_ids(pe, seen) for pe in self._all_inputs)
                                     and vb.on_chip_mix_rule(self._all_inputs))
        return self._windows

    def _fold(self, *words) -> None:
        """Tell the reducer what this rank is about to do and why (facts every rank must share): its agreement checks
        compare a hash of all of it across the ranks, so a rank that decides differently -- another pull sequence,
        another environment switch -- fails with "ranks out of step" instead of hanging the communicator."""
        fold = getattr(self._reducer, "fold", None)
        if fold is None:
            return
        if not self._folded_config:
            from . import voice_bank as vb
            self._folded_config = True
            fold(len(self._all_inputs), self._world, int(WINDOW_COLLECTIVES), int(bool(self._whole_windows())),
                 int(vb.BANK_WINDOWS), int(vb.BANK_WINDOWS_ANY_ROOT), int(vb.BANK_WINDOW_MAX_VOICES))
        fold(*words)

    def _render(self, start, duration):
        part = self._local.render(start, duration)
        if self._world == 1:
            return part
        if self._reducer is None:
            self._reducer = default_reducer()
        # only rows of a voice bank's window count as a window here (read-ahead / look-ahead rows are reduced block by
        # block: whether those layers opened a window is a rank-local matter)
        base = part._base if part._bank_window else None
        if base is None or not self._whole_windows():
            self._fold(1, start, duration)
            return self._reducer.all_reduce(part)
        # A row of the bank's window of 2, 4 or 8 blocks: the window is reduced whole when its first row is handed
        # out -- one collective of up to 1.5 MB instead of eight of 192 KB, each of which costs the links' latency --
        # and the blocks are rows of the reduced window.  Only a row that is USED orders the library stream behind
        # the collective; dropping one does not (the window keeps the buffers), so the next window is rendered while
        # this one is on the links, and the stream waits for it when the window after it replaces it.
        window, first_row = base
        if first_row == 0:
            self._fold(2, start, int(window.shape[0]))
            whole = self._reducer.all_reduce(Snippet(start, window))
            self._reduced = (window, whole, whole._dev)       # (the previous window's wait is enqueued here, when it is dropped)
        elif self._reduced is None or self._reduced[0] is not window:
            # a window whose first row did not pass through here (somebody pulled the local mix directly -- on every
            # rank, by the contract): its rows are reduced one by one
            self._fold(3, start, duration)
            return self._reducer.all_reduce(part)
        _, whole, buf = self._reduced

        def ready():
            whole._resolve()
        ready.on_use_only = True
        if buf is None:                                       # a reducer that answers on the host
            return Snippet(start, whole.data[first_row:first_row + duration])
        row = Snippet.window_rows(start, buf, first_row, duration)
        row._ready = ready
        return row

    def _on_start(self) -> None:
        self._reduced = None

    _on_stop = _on_start


# ----------------------------------------------------------------------------- bench workload
def c5_voice(pg, i: int):
    """BASELINE config 5 voice i: BlitSaw -> Biquad LP 2 kHz -> x ADSR(PeriodicGate)."""
    return pg.GainPE(
        pg.BiquadPE(pg.BlitSawPE(27.5 * 2 ** (i / 48.0)), frequency=2000.0, q=0.707),
        gain=pg.AdsrGatedPE(pg.PeriodicGate(2.0 + 0.01 * i, 0.5), 0.01, 0.1, 0.7, 0.2))


def c4_voice(pg, i: int, resonance: float = 0.3):
    """BASELINE config 4 instance i: 7-voice SuperSaw -> 24 dB ladder low-pass."""
    return pg.LadderPE(pg.SuperSawPE(55.0 * 2 ** (i / 12.0), voices=7, detune_cents=20.0, seed=i),
                       frequency=1200.0, resonance=resonance, mode=pg.LadderMode.LP24, drive=1.0, oversample=2)


def c4_res06_voice(pg, i: int):
    """C4 with the ladders above self-oscillation (resonance 0.6, as examples/17_ladder_filter.py:43): the warm-up of
    the time segments is found by trial (ladder_pe.SettleOptimist), the device check decides."""
    return c4_voice(pg, i, resonance=0.6)


def supersaw_voice(pg, i: int):
    """north_star's "512-voice SuperSaw mix": voice i is a 7-oscillator SuperSawPE."""
    return pg.SuperSawPE(55.0 * 2 ** (i / 96.0), voices=7, detune_cents=20.0, seed=i)


def mix_voice_factory(config: str):
    """(voice constructor, voice count) of the three sharded bench workloads."""
    return {"c5": (c5_voice, 512), "c4": (c4_voice, 64), "supersaw": (supersaw_voice, 512),
            "c4r06": (c4_res06_voice, 64)}[config]


def bench_voice_mix(pg, dist, steps, warmup, voices=512, block=48_000, config="c5"):
    """Sharded MixPE of BASELINE config 5 (512 voices), config 4 (64 SuperSaw->Ladder instances) or the
    512-voice SuperSaw mix: inputs i = rank (mod world) on each GPU, one RCCL all-reduce of the (block, 1)
    partial mix per rendered block.  Strong scaling.  `dist` offers world / rank / enabled / barrier() /
    max_over_ranks(); the communicator (pygmu2_amd.comm) is already up when world > 1."""
    import time

    from . import device

    pg.set_sample_rate(48000)
    world = dist.world if dist.enabled else 1
    rank = dist.rank if dist.enabled else 0
    make = mix_voice_factory(config)[0]
    parity = None
    if world > 1:
        # N-rank parity, the only kind a record of an N-GPU run can carry (the GPU test tier has one GPU): two blocks of
        # the sharded mix -- every rank takes part in the collectives -- against the unsharded MixPE of all voices
        # rendered by rank 0 alone.  Outside the timed region, on instances of their own.
        chk = ShardedMixPE([make(pg, i) for i in range(voices)], rank, world)
        rc = pg.NullRenderer(sample_rate=48000)
        rc.set_source(chk)
        rc.start()
        got = [chk.render(i * block, block).data.copy() for i in range(2)]
        rc.stop()
        if rank == 0:
            full = pg.MixPE(*[make(pg, i) for i in range(voices)])
            rf = pg.NullRenderer(sample_rate=48000)
            rf.set_source(full)
            rf.start()
            want = [full.render(i * block, block).data.copy() for i in range(2)]
            rf.stop()
            peak = max(float(np.max(np.abs(w))) for w in want)
            parity = max(float(np.max(np.abs(g.astype(np.float64) - w))) for g, w in zip(got, want)) / max(peak, 1e-30)
            del full, want
        del chk, got
        dist.barrier()
    root = ShardedMixPE([make(pg, i) for i in range(voices)], rank, world)
    r = pg.NullRenderer(sample_rate=48000)
    r.set_source(root)
    r.start()
    keep = {}
    for i in range(warmup):
        keep["s"] = root.render(i * block, block)
    keep["s"].dev
    device.synchronize()
    dist.barrier()
    calls0 = (getattr(root._reducer, "calls", 0), getattr(root._reducer, "floats", 0))
    t0 = time.perf_counter()
    for i in range(steps):
        keep["s"] = root.render((warmup + i) * block, block)
    keep["s"].dev                      # order the library stream behind the last block's all-reduce ...
    device.synchronize()               # ... and wait for it: every block is rendered AND reduced
    dist.barrier()
    dt = dist.max_over_ranks(time.perf_counter() - t0)
    calls1 = (getattr(root._reducer, "calls", 0), getattr(root._reducer, "floats", 0))
    # what a block costs this rank before the exchange: the same stream continued on the rank-local mix alone (no
    # collective), and with it what the blocks above spent waiting for the slowest rank and the links
    info = {"owned": len(root.owned)}
    if world > 1:                      # (a bank window of 2, 4, 8 blocks is one collective)
        info["collectives_in_timed_region"] = calls1[0] - calls0[0]
        info["floats_reduced_in_timed_region"] = calls1[1] - calls0[1]
        info["sharded_vs_unsharded_max_err_over_peak"] = parity
        checks = getattr(root._reducer, "checks", None)
        info["agreement_checks"] = checks() if checks else None
    probe = max(3, min(20, steps))
    pos = (warmup + steps) * block
    root.local.render(pos, block)
    device.synchronize()
    t1 = time.perf_counter()
    for i in range(probe):
        keep["s"] = root.local.render(pos + (1 + i) * block, block)
    keep["s"].dev
    device.synchronize()
    render = (time.perf_counter() - t1) / probe
    info["render_ms"] = round(render * 1e3, 4)
    info["render_ms_max"] = round(dist.max_over_ranks(render) * 1e3, 4)
    info["allreduce_wait_ms"] = round(max(0.0, dt / steps - dist.max_over_ranks(render)) * 1e3, 4) if world > 1 else 0.0
    r.stop()
    if config == "c5":
        name = (f"C5: {voices}-voice polyphonic graph (BlitSawPE->BiquadPE->xAdsrGatedPE per voice)->MixPE, "
                f"48 kHz mono, {block}-frame blocks, voices sharded i mod {world}")
    elif config == "supersaw":
        name = (f"{voices}-voice SuperSaw mix ({voices} x SuperSawPE 7 oscillators)->MixPE, 48 kHz mono, "
                f"{block}-frame blocks, voices sharded i mod {world}")
    else:
        name = (f"C4: {voices} x LadderPE(SuperSawPE 7 voices{', resonance 0.6' if config == 'c4r06' else ''})->MixPE, "
                f"48 kHz mono, {block}-frame blocks, instances sharded i mod {world}")
    return dt, block, name, info
