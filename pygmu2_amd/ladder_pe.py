"""
LadderPE: Moog-style 4-pole ladder with tanh feedback, oversampling and six responses
(ladder_pe.py:210-625).  The recurrence is nonlinear, so a chain advances sample by sample
in the reference's exact float64 operation order (pgx_ladder).  Independent chains (voices,
channels) fill the lanes of a wave; with scalar cutoff and resonance below self-oscillation
the block is additionally cut into time segments that warm up over `ladder_settle_frames`
samples -- verified on the device against the neighbouring segment, with a sequential
re-render of the chain if the check fails.
"""

from __future__ import annotations

from enum import Enum

import numpy as np

from . import device as _dev
from ._kernels import DeviceBuffer, check, lib, new_output, ptr
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


ACCURATE_TAIL_FACTOR = 4e-4
# At and above self-oscillation (k = 4 res 1.8 passes the small-signal loop gain of 4 at res 0.556; the reference's own
# example is there: examples/17_ladder_filter.py:43, res 0.6) and for loops that decay too slowly, ladder_settle_frames has
# no answer.  The device verifies every time segment against its neighbour and re-renders a chain sequentially when one
# disagrees, so a warm-up length only has to be TRIED (SettleOptimist): these, in turn.
OPTIMISTIC_SETTLES = (2048, 4096, 8192, 16384, 32768)
OPTIMISTIC_RETRY = 64                 # renders on the sequential kernel after every length failed, then the longest again
OPTIMISTIC = True
STATS = {"segmented": 0, "sequential": 0, "fallback_chains": 0, "repaired_samples": 0, "escalations": 0, "gave_up": 0}
SEGMENT_STREAM_LADDER = True          # PE-driven cutoff / resonance: time segments too (warm-up from the block's range)
STREAM_SEGMENT_MIN_FRAMES = 8192      # shorter blocks stay on the sequential kernel (the range costs a read-back)


class LadderMode(Enum):
    LP24 = "lp24"
    LP12 = "lp12"
    BP24 = "bp24"
    BP12 = "bp12"
    HP24 = "hp24"
    HP12 = "hp12"


_MODE_INDEX = {m: i for i, m in enumerate(LadderMode)}


def ladder_settle_frames(cutoff: float, resonance: float, sample_rate: float, oversample: int,
                         limit: int = 16384, target: float = 2e-10) -> int:
    """
    Samples after which the ladder has forgotten its state to `target`: the smallest W for which
    every entry of M^(W*oversample) is below it, M being the small-signal (tanh' = 1) transition
    matrix of one sub-step over the states (z0[4], z1[4]) of ladder_pe.py:139-181 -- saturation only
    lowers the loop gain, so the small-signal loop is the slowest to forget.  0 = do not segment
    (at or above self-oscillation, or too slow a decay).  The device checks every segment to 1e-8 and
    re-renders on failure, so this only has to be a good estimate; a block's run time is proportional
    to it.  target = 2e-10: with 8 states of magnitude <= 2 the carried-over difference stays below
    8 * 2e-10 * 2 = 3.2e-9 -- a third of what the device check tolerates, 1/20 of a float32 ulp at full scale.
    """
    nyquist = sample_rate / 2.0
    fc = min(max(float(cutoff), 5.0), min(nyquist * 0.85, nyquist - 1.0))
    wc = fc * 2.0 * np.pi / (sample_rate * oversample)
    alpha = 0.9892 * wc - 0.4324 * wc ** 2 + 0.1381 * wc ** 3 - 0.0202 * wc ** 4
    q_adjust = 1.006 + 0.0536 * wc - 0.095 * wc ** 2 - 0.05 * wc ** 4
    k_fb = 4.0 * min(max(float(resonance), 0.0), 1.0) * 1.8 * q_adjust
    c0, c1 = 0.76923077, 0.23076923
    # rows: new z0[0..3], new z1[0..3]; columns: old z0[0..3], old z1[0..3]
    m = np.zeros((8, 8))
    u = np.zeros(8)
    u[7] = -k_fb                                   # stage input of stage 0: -(k*q_adjust) * z1[3]
    stage_in = u
    for j in range(4):
        ft = alpha * c0 * stage_in
        ft[j] += alpha * c1
        ft[4 + j] += 1.0 - alpha
        m[j] = stage_in                             # z0[j]' = this stage's input
        m[4 + j] = ft                               # z1[j]' = its output
        stage_in = ft.copy()
    powers = [m]                                    # M^(2^k)
    with np.errstate(over="ignore", invalid="ignore"):
        while np.max(np.abs(powers[-1])) >= target:
            if len(powers) > 18 or not np.all(np.isfinite(powers[-1])):
                return 0
            powers.append(powers[-1] @ powers[-1])
        # smallest exponent: greedy over the bits below the first power that is already small enough
        acc, steps = np.eye(8), 0
        for k in range(len(powers) - 2, -1, -1):
            trial = acc @ powers[k]
            if np.max(np.abs(trial)) >= target:
                acc, steps = trial, steps + (1 << k)
        steps += 1
    frames = (-(-steps // oversample) + 31) // 32 * 32
    return frames if frames <= limit else 0


class SettleOptimist:
    """Warm-up lengths by trial, for ladders whose small-signal loop does not decay (or decays too slowly for
    ladder_settle_frames).  A driven, saturating ladder usually locks to its input: two trajectories under the same
    input then converge although the linearised loop oscillates (measured on the CPU oracle, experiments/README.md: a
    110 Hz saw, a SuperSaw, noise or a chord into an 800 Hz ladder at resonance 0.6 ... 1.0 forget a zero start to 1e-8
    within 2 000 ... 8 000 samples; a lone sine does not at resonance >= 0.7 -- the loop runs free beside it).  The
    device check (k_ladder_finish: every segment's entry state against its neighbour's exit state, 1e-8) decides; this
    class only reads its verdict back -- the cumulative fallback counter at the head of the workspace, copied
    asynchronously -- and moves to the next length, or to the sequential kernel for `OPTIMISTIC_RETRY` renders
    (doubling, at most 1024) when the longest failed too.  A failed trial costs the sequential re-render of the chain
    (what the render would have cost without the trial), so the host must not run ahead of the verdicts: a length
    that has not passed yet is waited for before the next render is planned (one device wait per new length), a
    length that has passed may have `RUN_AHEAD` verdicts outstanding."""

    RUN_AHEAD = 2

    def __init__(self):
        self.level = 0
        self.sleep = 0
        self.retry = OPTIMISTIC_RETRY
        self.seen = 0                  # the counters' values so far: chains with a repair, samples repaired
        self.seen_repaired = 0
        self.pending = []              # (ticket, pinned view, level, workspace kept alive)
        self.good = 0                  # verified renders in a row at this level

    def settle(self, frames: int) -> tuple[int, int]:
        """(settle_frames, accurate_frames) for the next render of `frames` frames; (0, 0): sequential."""
        self.poll()
        while len(self.pending) > (self.RUN_AHEAD if self.good > 0 else 0):
            self.poll(wait=True, only_first=True)
        if not OPTIMISTIC:
            return 0, 0
        if self.sleep > 0:
            self.sleep -= 1
            return 0, 0
        w = OPTIMISTIC_SETTLES[self.level]
        return w, w // 2               # (the float32-tanh half leaves ~1e-7, the accurate half contracts it)

    def launched(self, workspace, level_settle: int, frames: int = 0, chains: int = 1) -> None:
        """A segmented render with this warm-up has been enqueued: its verdict is read back without waiting."""
        STATS["segmented"] += 1
        view, ticket = workspace.rows(0, 16).begin_to_host()
        self.pending.append((ticket, view, level_settle, workspace, int(frames), max(1, int(chains))))

    def poll(self, wait: bool = False, only_first: bool = False) -> None:
        first = True
        while self.pending and (first or not only_first):
            first = False
            ticket, view, settle, _ws, frames, chains = self.pending[0]
            if wait:
                _dev.wait_to_host(ticket)
            elif not _dev.host_copy_done(ticket):
                return
            self.pending.pop(0)
            count = int(view.view(np.int32)[0])
            repaired = int(view.view(np.int64)[1])
            if count > self.seen:
                STATS["fallback_chains"] += count - self.seen
                self.seen = count
            redone, self.seen_repaired = repaired - self.seen_repaired, repaired
            STATS["repaired_samples"] += redone
            # The device repairs what disagrees (k_ladder_finish: from the first bad boundary until the trajectory
            # meets a segment's own entry state again), one lane per chain at ~0.5 us per sample; a lane's warm-up is
            # ~0.18 us per sample.  A longer warm-up pays when the repairs of a launch cost more than the warm-up does.
            if redone * 3 > settle * max(1, chains // 4):
                self.good = 0
                if self.sleep == 0 and settle == OPTIMISTIC_SETTLES[self.level]:     # (not escalated since)
                    if self.level + 1 < len(OPTIMISTIC_SETTLES):
                        self.level += 1
                        STATS["escalations"] += 1
                    elif frames and redone * 2 > frames * chains:
                        # the longest warm-up, and still most of the render was repaired: a ladder that runs free
                        # beside its input.  The sequential kernel for a while, then the longest length again.
                        self.sleep = self.retry
                        self.retry = min(self.retry * 2, 1024)
                        STATS["gave_up"] += 1
            else:
                self.good += 1

    def reset(self) -> None:
        self.poll()


class LadderPE(ProcessingElement):
    _LOOK_AHEAD_SAFE = True            # look_ahead.py
    _STATE_FIELDS = ("_state", "_state_channels")

    _DEFAULT_OVERSAMPLE = 2
    _RESONANCE_MULTIPLIER = 1.8
    _MIN_CUTOFF_FREQ = 5.0
    _STATE_DECAY = 0.95
    _INPUT_THRESHOLD = 1e-5

    def __init__(self, source: ProcessingElement, frequency, resonance=0.0,
                 mode: LadderMode = LadderMode.LP24, drive=1.0, passband_gain: float = 0.5,
                 oversample: int = _DEFAULT_OVERSAMPLE):
        self._source = source
        self._frequency = frequency
        self._resonance = resonance
        self._mode = mode
        self._drive = drive
        self._passband_gain = float(np.clip(passband_gain, 0.0, 0.5))
        self._oversample = max(1, int(oversample))
        self._freq_is_pe = isinstance(frequency, ProcessingElement)
        self._res_is_pe = isinstance(resonance, ProcessingElement)
        self._drive_is_pe = isinstance(drive, ProcessingElement)
        self._params: DeviceBuffer | None = None
        self._state: DeviceBuffer | None = None      # [C][9]: z0[4], z1[4], old_input
        self._state_channels = 0
        self._workspace: DeviceBuffer | None = None
        self._range_dev: DeviceBuffer | None = None          # (2, 256, 2) float64: min / max pairs of the two control streams
        self._range_pending: list = []
        self._range_seen: list = []
        self._stream_settle_cache: dict = {}
        self._stream_accurate = 0
        self._optimist: SettleOptimist | None = None        # warm-up lengths by trial (no analytic estimate)

    source = property(lambda self: self._source)
    frequency = property(lambda self: self._frequency)
    resonance = property(lambda self: self._resonance)
    mode = property(lambda self: self._mode)
    drive = property(lambda self: self._drive)
    passband_gain = property(lambda self: self._passband_gain)
    oversample = property(lambda self: self._oversample)

    def inputs(self) -> list[ProcessingElement]:
        out = [self._source]
        for is_pe, p in ((self._freq_is_pe, self._frequency), (self._res_is_pe, self._resonance),
                         (self._drive_is_pe, self._drive)):
            if is_pe:
                out.append(p)
        return out

    def is_pure(self) -> bool:
        return False

    def channel_count(self) -> int | None:
        return self._source.channel_count()

    def _compute_extent(self) -> Extent:
        ext = self._source.extent()
        for is_pe, p in ((self._freq_is_pe, self._frequency), (self._res_is_pe, self._resonance),
                         (self._drive_is_pe, self._drive)):
            if is_pe:
                ext = ext.intersection(p.extent())      # strict: no fallback (ladder_pe.py:330-346)
        return ext

    def _reset_state(self) -> None:
        if self._state is not None:
            self._state.zero_()
        self._range_pending, self._range_seen = [], []       # a new stream: its first block waits for its own range

    _on_start = _reset_state
    _on_stop = _reset_state

    def _scalar_params(self) -> dict:
        def scalar(is_pe, p):
            return 0.0 if is_pe else float(p)
        return dict(freq=scalar(self._freq_is_pe, self._frequency),
                    resonance=scalar(self._res_is_pe, self._resonance),
                    drive=scalar(self._drive_is_pe, self._drive),
                    passband_gain=self._passband_gain, oversample=self._oversample,
                    mode=_MODE_INDEX[self._mode])

    def _render(self, start: int, duration: int) -> Snippet:
        src = self._source.render(start, duration)
        ch = src.channels
        if self._state is None or self._state_channels != ch:
            self._state = DeviceBuffer((ch, 9), np.float64, zero=True)
            self._state_channels = ch
        if self._params is None:
            self._params = _dev.upload_struct(_dev.LADDER_PARAMS, **self._scalar_params())
        _, f_buf = self._control_stream(self._frequency, start, duration)
        _, r_buf = self._control_stream(self._resonance, start, duration)
        _, d_buf = self._control_stream(self._drive, start, duration)
        out = new_output(duration, ch)
        settle = self._settle_frames()
        streams = self._freq_is_pe or self._res_is_pe
        if streams and SEGMENT_STREAM_LADDER:
            settle = self._settle_frames_for_streams(f_buf, r_buf, duration)
        accurate = self._stream_accurate if streams else self._accurate_frames()
        trial = False
        if settle == 0 and duration >= STREAM_SEGMENT_MIN_FRAMES and (not streams or SEGMENT_STREAM_LADDER):
            if self._optimist is None:
                self._optimist = SettleOptimist()
            settle, accurate = self._optimist.settle(duration)
            trial = settle > 0
        L = lib()
        need = L.pgx_ladder_workspace_bytes(1, duration, ch, settle)
        if need and (self._workspace is None or self._workspace.nbytes < need):
            counters = None if self._workspace is None else self._workspace.rows(0, 16)
            self._workspace = DeviceBuffer((need,), np.uint8, zero=True)
            if counters is not None:                 # the counters are cumulative: they move with the workspace
                check(L.pgx_memcpy_d2d(self._workspace.ptr, counters.ptr, 16), "pgx_memcpy_d2d")
        check(L.pgx_ladder(out.ptr, 0, src.dev.ptr, 0, 1, duration, ch, float(self.sample_rate),
                           self._params.ptr, ptr(f_buf), ptr(r_buf), ptr(d_buf), self._state.ptr, settle,
                           accurate, ptr(self._workspace) if need else None), "pgx_ladder")
        if trial and need:
            self._optimist.launched(self._workspace, settle, duration, ch)
        elif not need:
            STATS["sequential"] += 1
        return Snippet(start, out)

    def _settle_frames(self) -> int:
        if self._freq_is_pe or self._res_is_pe:
            return 0
        return ladder_settle_frames(self._frequency, self._resonance, self.sample_rate, self._oversample)

    def _stream_ranges(self, f_buf, r_buf, duration: int):
        """(lowest cutoff, highest resonance) the warm-up is planned for, None where the control is a scalar.
        The block's own range is asked for (one small launch per control stream, one asynchronous read-back of both)
        but not waited for: a device wait per block would undo the pipelining of everything queued before it.  The
        plan uses the ranges that HAVE arrived -- the previous blocks' -- extended by their trend (a sweep keeps
        moving), and only a stream's first block waits.  A stale estimate cannot hurt the result: the device
        verifies every segment against its neighbour and re-renders the chain when one disagrees."""
        parts = int(min(256, max(1, duration // 4096)))
        if self._range_dev is None:
            self._range_dev = DeviceBuffer((2, 256, 2), np.float64)
            self._range_pending = []                 # (ticket, host view, parts, has f, has r)
            self._range_seen = []                    # the last two (lo_f, hi_r) that arrived
        L = lib()
        if self._range_pending:                      # the buffer is written again: behind the copy that still reads it
            _dev.fence_to_host(self._range_pending[-1][0])
        if f_buf is not None:
            check(L.pgx_stream_range(self._range_dev.ptr, parts, f_buf.ptr, duration), "pgx_stream_range")
        if r_buf is not None:
            check(L.pgx_stream_range(self._range_dev.offset_ptr(512), parts, r_buf.ptr, duration), "pgx_stream_range")
        view, ticket = self._range_dev.begin_to_host()
        self._range_pending.append((ticket, view, parts, f_buf is not None, r_buf is not None))
        while self._range_pending:
            ticket, view, p, has_f, has_r = self._range_pending[0]
            if not self._range_seen and len(self._range_pending) == 1:
                _dev.wait_to_host(ticket)            # the stream's first block: nothing to go by yet
            elif not _dev.host_copy_done(ticket):
                break
            self._range_pending.pop(0)
            lo_f = float(np.min(view[0, :p, 0])) if has_f else None
            hi_r = float(np.max(view[1, :p, 1])) if has_r else None
            self._range_seen = (self._range_seen + [(lo_f, hi_r)])[-2:]
        last = self._range_seen[-1]
        prev = self._range_seen[-2] if len(self._range_seen) > 1 else last
        behind = 1 + len(self._range_pending)        # blocks between the newest range and the block being planned
        lo_f = hi_r = None
        if last[0] is not None:
            lo_f = last[0] - behind * max(0.0, (prev[0] if prev[0] is not None else last[0]) - last[0])
        if last[1] is not None:
            hi_r = last[1] + behind * max(0.0, last[1] - (prev[1] if prev[1] is not None else last[1]))
        return lo_f, hi_r

    def _settle_frames_for_streams(self, f_buf, r_buf, duration: int) -> int:
        """Warm-up length of the time segments when cutoff and / or resonance are PEs: the ladder forgets slowest at
        the lowest cutoff and the highest resonance it sees, so the estimate of ladder_settle_frames is taken there
        (quantised so that a sweep does not recompute it every block, and half as much again for the coefficients
        moving under the warm-up).  Only an estimate is needed: the device verifies every segment against its
        neighbour and re-renders the chain sequentially if one disagrees."""
        if duration < STREAM_SEGMENT_MIN_FRAMES:
            return 0
        lo_f, hi_r = self._stream_ranges(f_buf, r_buf, duration)
        cutoff = float(self._frequency) if lo_f is None else lo_f
        res = float(self._resonance) if hi_r is None else hi_r
        if not (np.isfinite(cutoff) and np.isfinite(res)):
            return 0
        cutoff = max(cutoff, 5.0)
        key = (int(np.floor(np.log2(cutoff) * 8.0)), int(np.ceil(min(max(res, 0.0), 1.0) * 50.0)))
        hit = self._stream_settle_cache.get(key)
        if hit is None:
            fc, rs = 2.0 ** (key[0] / 8.0), key[1] / 50.0
            est = ladder_settle_frames(fc, rs, self.sample_rate, self._oversample)
            tail = ladder_settle_frames(fc, rs, self.sample_rate, self._oversample, target=ACCURATE_TAIL_FACTOR)
            hit = self._stream_settle_cache[key] = (int(est * 1.5), int(tail * 1.5)) if est else (0, 0)
        self._stream_accurate = hit[1]
        return hit[0]

    def _accurate_frames(self) -> int:
        """Tail of a segment's warm-up that needs the float64 tanh: long enough to contract the ~1e-7 the
        float32-tanh part leaves in the state down to the warm-up target (2e-10): a factor of 4e-4."""
        if self._freq_is_pe or self._res_is_pe:
            return 0
        return ladder_settle_frames(self._frequency, self._resonance, self.sample_rate, self._oversample,
                                    target=ACCURATE_TAIL_FACTOR)

    def __repr__(self) -> str:
        def s(is_pe, p):
            return f"{type(p).__name__}(...)" if is_pe else p
        return (f"LadderPE(source={type(self._source).__name__}, "
                f"frequency={s(self._freq_is_pe, self._frequency)}, "
                f"resonance={s(self._res_is_pe, self._resonance)}, mode={self._mode.value}, "
                f"drive={s(self._drive_is_pe, self._drive)}, oversample={self._oversample})")
