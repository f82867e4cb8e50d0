"""
Voice bank: batched rendering of structurally identical voice sub-graphs under a MixPE.

The reference renders a MixPE of K voices by pulling K separate sub-graphs, one numpy
call chain each (mix_pe.py:69-96).  On MI355X K x (PEs per voice) small launches would
leave the chip idle, so when every input of a MixPE is the same tree of supported PEs
with scalar parameters, the trees are walked in lock-step and each level is rendered by
ONE batched launch: per-voice parameter blocks and state blobs are stacked [K] in HBM,
intermediate signals are [K][frames][channels], and the final pgx_mix_batch adds the
K voices in input order in float32 -- the same additions, in the same order, as MixPE.
The batched kernels are the very same kernels the single PEs use (batch = 1 there), so a
bank produces the same samples as rendering the voices one by one.

Supported nodes: SinePE (scalar), BlitSawPE (scalar), SuperSawPE (scalar), BiquadPE
(constant coefficients), LadderPE (scalar controls), CombPE (scalar controls), GainPE (constant or PE gain),
AdsrGatedPE, PeriodicGate.  Anything else -> `try_build_bank` returns None and MixPE
falls back to per-input rendering.
"""

from __future__ import annotations

import os

import numpy as np

from . import device as _dev
from ._kernels import DeviceBuffer, blitsaw_workspace, check, lib, ptr
from .adsr_pe import AdsrGatedPE
from .biquad_pe import BiquadPE, rbj_coefficients, settle_frames, settle_frames_fine
from .blit_saw_pe import BlitSawPE, wide_oscillators_ok
from .comb_pe import CombPE
from .extent import Extent
from .gain_pe import GainPE
from .ladder_pe import LadderPE
from .periodic_gate import PeriodicGate
from .processing_element import ProcessingElement
from .sine_pe import SinePE
from .snippet import Snippet
from .super_saw_pe import SuperSawPE

MIN_VOICES = 4
BANK_WINDOWS = True           # a small bank streamed in equal blocks: 2, 4, 8 blocks per render, handed out as rows (VoiceBank)
BANK_WINDOWS_ANY_ROOT = os.environ.get("PGX_BANK_WINDOWS_ANY_ROOT", "0") == "1"     # experiments (C5: measured slower)
BANK_WINDOW_FIRST = 2
BANK_WINDOW_MAX = int(os.environ.get("PGX_BANK_WINDOW_MAX", "8"))
BANK_WINDOW_FRAMES = 1 << 20
BANK_WINDOW_MAX_VOICES = int(os.environ.get("PGX_BANK_WINDOW_MAX_VOICES", "256"))  # (a bank that fills the chip gains nothing: its launches are long)
LADDER_WINDOWS = True         # a ladder bank directly under the mix, streamed in equal blocks: several blocks per launch
LADDER_WINDOW_FIRST = 2       # ... 2, then 4, 8, 16, 32 blocks (a lane of k_ladder_segments pays its 1024-sample warm-up once per
                              # launch: C4 over a stream 0.0675 ms per block with windows up to 8, 0.0591 up to 16, 0.0497 up to 32
                              # -- tools/c4_windows_probe.py; the same ladders at resonance 0.6, whose warm-up is 2048: 0.226 /
                              # 0.134 / 0.088; like look-ahead's windows a stream that stops after k blocks has rendered < 2k + 2)
LADDER_WINDOW_MAX = int(os.environ.get("PGX_LADDER_WINDOW_MAX", "32"))
LADDER_WINDOW_FRAMES = int(os.environ.get("PGX_LADDER_WINDOW_FRAMES", str(1 << 21)))   # ... and at most this many frames
PREFETCH_LADDER_INPUT = True
PREFETCH_SUPERSAW_VOICES = True   # small SuperSaw banks under the mix: oscillators one block ahead (VoiceBank._supersaw_pipelined)  # _LadderNode: next block's oscillators beside this block's ladder
FUSED_VOICE_MIN = 128        # voices (one workgroup each) from which BlitSaw -> Biquad runs as one launch
SEGMENTED_CHAIN_MAX = int(os.environ.get("PGX_BBW_MAX_BATCH", "256"))
SEGMENTED_CHAIN = True       # ... and smaller banks (4 .. 256 voices: a rank's share of C5) as one launch in concurrent time
                             # segments (pgx_blitsaw_biquad_wide_seg: closed-form oscillator carries, the filters warm up)
PIPELINE_FULL_SUPERSAW_BANK = True    # ... and for a bank that fills the chip (512 instances: the 17 us mix beside the next block's bank)
PIPELINE_SUPERSAW_BANK = False  # the same overlap for the voices-summed-on-chip bank: measured slower (render_mix)
VOICE_TILES = os.environ.get("PGX_VOICE_TILES", "1") != "0"   # BlitSaw -> Biquad [-> x envelope] voices mixed on chip (pgx_voice_tiles)
VOICE_TILES_MIN_FRAMES = 4096
VOICE_TILES_MAX_WARM = 2048  # pgx_voice_tiles_max_warm(): a filter has to settle (every entry of A^warm below 2^-90) within this many frames
VOICE_TILE_WINDOWS = os.environ.get("PGX_VOICE_TILE_WINDOWS", "1") != "0"    # ... in windows of 2, 4, 8 blocks: the stream synchronisations,
                             # the edge searches and the small launches of a block are per window then, and the envelopes of the NEXT
                             # window are walked beside this one's voices (C5: 94 -> 83 us per block)
VOICE_TILES_MIN_VOICES = int(os.environ.get("PGX_VOICE_TILES_MIN_VOICES", "16"))   # (192 while the mix ran block by block: the
                             # envelope walk, which the on-chip mix waits for BEFORE the voices, was then the block's critical chain for
                             # smaller banks -- 64 voices 56 us per block against 52.  In windows: 8 voices 33 against 45, 64 voices 40
                             # against 49, 128: 48 against 56, 256: 63 against 72, 512: 83 against 106.  Smaller banks keep the kernels
                             # that are bit-identical to the per-voice path)
FUSE_GAIN_IN_CHAIN = False   # ... and multiplied into the voices by the oscillator -> filter kernel: measured, no gain (render_mix)
EARLY_WALK_MAX_VOICES = int(os.environ.get("PGX_EARLY_WALK_MAX", "256"))  # ... started at once (not behind the block's oscillators) for banks up to this size
ENVELOPE_AHEAD = True        # a bank's AdsrGatedPE(PeriodicGate) envelopes one block ahead on the side stream (render_mix)
WIDE_SUPERSAW = True         # the bank kernel with 16 frames per thread (pgx_supersaw_wide) where its conditions hold
SEGMENTED_SUPERSAW = True    # below FUSED_SUPERSAW_MIN: the fused bank in concurrent time segments (closed-form carries)
FUSED_SUPERSAW_MIN = 257     # SuperSaw instances from which the bank is rendered in one segment per instance, one block ahead (256 -- a rank's
                             # share at two ranks -- is better off in two time segments and windows: 97 -> 84 us per block)


def _is_pe(x) -> bool:
    return isinstance(x, ProcessingElement)


class _Rows:
    """Frames [offset, offset + n) of every voice of a [K][total][channels] buffer: what a node that renders several
    blocks at once hands out per block (`.ptr` at the first voice's first frame, `.stride` elements from voice to voice)."""
    __slots__ = ("buf", "ptr", "shape", "stride")

    def __init__(self, buf, offset: int, n: int):
        k, total, ch = buf.shape
        self.buf = buf
        self.ptr = buf.offset_ptr(offset * ch)
        self.shape = (k, n, ch)
        self.stride = total * ch


class _Node:
    """One level of the voice tree: K PE instances of the same class and static config."""

    def __init__(self, pes, children):
        self.pes = pes
        self.k = len(pes)
        self.children = children
        self.sr = float(pes[0].sample_rate)

    def reset(self):
        for c in self.children.values():
            c.reset()

    # ---- what VoiceBank's windows need of a node: its carried state by name, and a way to come to rest
    _STATE = ()                  # attributes (DeviceBuffers, numbers, None) that make up the carried state

    def quiesce(self, keep=None) -> None:
        """Anything rendered ahead of the stream is dropped: the states are where the last block handed out left them."""

    def snapshot(self) -> dict:
        snap = {}
        for name in self._STATE:
            value = getattr(self, name)
            if isinstance(value, DeviceBuffer):
                twin = DeviceBuffer(value.shape, value.dtype)
                check(lib().pgx_memcpy_d2d(twin.ptr, value.ptr, value.nbytes), "pgx_memcpy_d2d")
                value = twin
            snap[name] = value
        return snap

    def restore(self, snap: dict) -> None:
        for name, value in snap.items():
            mine = getattr(self, name)
            if isinstance(value, DeviceBuffer) and isinstance(mine, DeviceBuffer) and mine.shape == value.shape:
                check(lib().pgx_memcpy_d2d(mine.ptr, value.ptr, value.nbytes), "pgx_memcpy_d2d")
            else:
                setattr(self, name, value)

    def channels(self) -> int:
        raise NotImplementedError

    def render(self, start: int, n: int) -> DeviceBuffer:
        """-> DeviceBuffer [K][n][channels] float32."""
        raise NotImplementedError


class _SineNode(_Node):
    def __init__(self, pes):
        super().__init__(pes, {})
        rec = np.zeros(self.k, dtype=_dev.SINE_PARAMS)
        for i, pe in enumerate(pes):
            rec[i] = (2.0 * np.pi * float(pe._frequency), float(pe._amplitude), float(pe._phase))
        self.params = _dev.upload_structs(rec)
        self.ch = pes[0]._channels

    def channels(self):
        return self.ch

    def render(self, start, n):
        out = DeviceBuffer((self.k, n, self.ch), np.float32)
        check(lib().pgx_sine_render(out.ptr, n * self.ch, self.k, start, n, self.ch, self.sr,
                                    self.params.ptr), "pgx_sine_render")
        return out


class _BlitSawNode(_Node):
    _STATE = ("state", "last_end")

    def __init__(self, pes):
        super().__init__(pes, {})
        rec = np.zeros(self.k, dtype=_dev.BLITSAW_PARAMS)
        for i, pe in enumerate(pes):
            for key, v in pe._scalar_params().items():
                rec[i][key] = v
        self.params = _dev.upload_structs(rec)
        self.init_state = np.stack([pe._initial_state() for pe in pes])
        self.state = DeviceBuffer((self.k, 2), np.float64)
        self.ch = pes[0]._channels
        self.last_end = None
        # a few oscillators in concurrent time segments: the SuperSaw bank kernel with one voice per instance
        # (_SuperSawNode.segmented); needs the automatic (odd) M and a leak below 1 for the closed-form carries
        self.closed_form_ok = bool(np.all(rec["m"] < 0.0) and np.all(rec["leak"] > 0.0)
                                   and np.all(rec["leak"] <= 0.9999) and np.all(rec["freq"] >= 1.0))
        self.wide_ok = wide_oscillators_ok(rec, self.sr)           # (_SuperSawNode.wide)
        self.state_alt = None
        self.tables = {}
        self.unit_amp = None

    def reset(self):
        self.last_end = None

    def channels(self):
        return self.ch

    def prepare(self, start):
        """The reset rule of blit_saw_pe.py: a render that does not continue the previous one starts over."""
        if self.last_end is None or start != self.last_end:
            self.state.upload(self.init_state)

    def wide(self) -> bool:
        return WIDE_SUPERSAW and self.wide_ok

    def segmented(self, n: int) -> bool:
        L = lib()
        segments = (L.pgx_supersaw_wide_segments(self.k, 1, n) if self.wide()
                    else L.pgx_supersaw_bank_segments(self.k, n))
        return SEGMENTED_SUPERSAW and self.closed_form_ok and 4 <= self.k < FUSED_SUPERSAW_MIN and segments > 1

    def _render_segments(self, start, n):
        """The SuperSaw bank kernels with one voice per instance and unit instance amplitude: float32(y * 2 amp * 1.0)
        is the oscillator's own sample (pgx_supersaw_wide; pgx_supersaw_bank_seg rounds to float32 twice, the same)."""
        L = lib()
        self.prepare(start)
        if self.state_alt is None:
            self.state_alt = DeviceBuffer(self.state.shape, self.state.dtype)
        if self.unit_amp is None:
            self.unit_amp = DeviceBuffer.from_host(np.ones(self.k, dtype=np.float64))
        kind = "wide" if self.wide() else "bank"
        tables = self.tables.get(kind)
        if tables is None:
            size, make = ((L.pgx_supersaw_wide_table_bytes, L.pgx_supersaw_wide_tables) if kind == "wide"
                          else (L.pgx_supersaw_bank_table_bytes, L.pgx_supersaw_bank_tables))
            tables = self.tables[kind] = DeviceBuffer((size(self.k, 1),), np.uint8)
            check(make(tables.ptr, self.k, 1, self.sr, self.params.ptr), f"pgx_supersaw_{kind}_tables")
        out = DeviceBuffer((self.k, n, self.ch), np.float32)
        if kind == "wide":
            check(L.pgx_supersaw_wide(out.ptr, n * self.ch, self.k, 1, n, self.ch, self.state.ptr,
                                      self.state_alt.ptr, self.unit_amp.ptr, tables.ptr), "pgx_supersaw_wide")
        else:
            check(L.pgx_supersaw_bank_seg(out.ptr, n * self.ch, self.k, 1, n, self.ch, self.sr, self.params.ptr,
                                          self.state.ptr, self.state_alt.ptr, self.unit_amp.ptr, tables.ptr),
                  "pgx_supersaw_bank_seg")
        self.state, self.state_alt = self.state_alt, self.state
        self.last_end = start + n
        return out

    def render(self, start, n):
        if self.segmented(n):
            return self._render_segments(start, n)
        self.prepare(start)
        out = DeviceBuffer((self.k, n, self.ch), np.float32)
        ws = blitsaw_workspace(self, self.k, n, False)
        check(lib().pgx_blitsaw(out.ptr, n * self.ch, self.k, n, self.ch, self.sr, self.params.ptr,
                                None, 0, None, 0, None, 0, self.state.ptr, ptr(ws), None), "pgx_blitsaw")
        self.last_end = start + n
        return out


class _SuperSawNode(_Node):
    """Bank of SuperSawPEs.  From FUSED_SUPERSAW_MIN instances on: one launch, voices summed on chip.  Below
    (a rank's share of a sharded mix): the oscillators of all instances in one launch (`_voices`), then the
    ordered voice sum.  When the node feeds the bank's mix directly VoiceBank pipelines the two (see
    VoiceBank._supersaw_pipelined): `ahead` then holds the oscillator samples of the block after the last one."""

    _STATE = ("state", "last_end")

    def quiesce(self, keep=None):
        self._forget_ahead(restore=True)
        if self.ahead_bank is not None:
            self._forget_bank_ahead(restore=True)

    def __init__(self, pes):
        super().__init__(pes, {})
        self.nv = len(pes[0]._oscillators)
        self.params = _dev.upload_structs(np.concatenate([pe._voice_param_records() for pe in pes]))
        self.init_state = np.concatenate([pe._voice_initial_state() for pe in pes])
        self.state = DeviceBuffer((self.k * self.nv, 2), np.float64)
        self.amp = DeviceBuffer.from_host(np.array([float(pe._amplitude) for pe in pes], dtype=np.float64))
        self.ch = pes[0]._channels
        self.last_end = None
        self.ahead = None            # (start, n, voices buffer, (state copy, last_end))
        rec = np.concatenate([pe._voice_param_records() for pe in pes])
        self.closed_form_ok = bool(np.all(rec["m"] < 0.0) and np.all(rec["leak"] > 0.0)
                                   and np.all(rec["leak"] <= 0.9999) and np.all(rec["freq"] >= 1.0))
        # pgx_supersaw_wide (16 frames per thread): the rotation / recurrence form of the Dirichlet kernel only -- scalar
        # frequency, the automatic M -- and the closed-form carries of the time segments (blit_saw_pe.wide_oscillators_ok)
        self.wide_ok = wide_oscillators_ok(rec, self.sr)
        self.state_alt = None        # the segmented bank reads one state buffer and writes the other
        self.ahead_bank = None       # (start, n, bank output, last_end before): VoiceBank._supersaw_pipelined
        self.tables = {}             # ... and loads what depends on the parameters only (pgx_supersaw_*_tables)

    def fused(self) -> bool:
        return self.k >= FUSED_SUPERSAW_MIN and self.nv <= 16

    def wide(self) -> bool:
        return WIDE_SUPERSAW and self.wide_ok and self.nv <= 16

    def segmented(self, n: int) -> bool:
        """Fewer instances than fill the chip (a rank's share of a sharded mix): the fused bank in concurrent time
        segments (pgx_supersaw_wide / pgx_supersaw_bank_seg), carries from the integrator's closed form -- automatic
        (odd) M, leak < 1."""
        L = lib()
        segments = (L.pgx_supersaw_wide_segments(self.k, self.nv, n) if self.wide()
                    else L.pgx_supersaw_bank_segments(self.k, n))
        return (SEGMENTED_SUPERSAW and not self.fused() and self.nv <= 16 and self.closed_form_ok
                and segments > 1)

    def _forget_ahead(self, restore: bool) -> None:
        ahead, self.ahead = self.ahead, None
        if ahead is not None and restore:
            saved, last_end = ahead[3]
            check(lib().pgx_memcpy_d2d(self.state.ptr, saved.ptr, saved.nbytes), "pgx_memcpy_d2d")
            self.last_end = last_end

    def reset(self):
        self._forget_ahead(restore=False)
        self.ahead_bank = None
        self.last_end = None

    def channels(self):
        return self.ch

    def _voices(self, start, n, backup=None):
        """[instances * voices][n] float32 oscillator samples, on the current stream.  backup: a buffer that
        receives the states on entry (written by the oscillator kernel itself: no extra launch)."""
        if self.ahead_bank is not None:
            self._forget_bank_ahead(restore=True)
        if self.last_end is None or start != self.last_end:
            self.state.upload(self.init_state)
        voices = DeviceBuffer((self.k * self.nv, n), np.float32)
        ws = blitsaw_workspace(self, self.k * self.nv, n, False)
        check(lib().pgx_blitsaw(voices.ptr, n, self.k * self.nv, n, 1, self.sr, self.params.ptr,
                                None, 0, None, 0, None, 0, self.state.ptr, ptr(ws), ptr(backup)), "pgx_blitsaw")
        self.last_end = start + n
        return voices

    def take_voices(self, start, n):
        """The oscillator samples of (start, n): the ones rendered ahead if they are these, else rendered now
        (after the states went back to where the caller's last block left them)."""
        if self.ahead is not None:
            if self.ahead[0] == start and self.ahead[1] == n:
                voices, self.ahead = self.ahead[2], None
                return voices
            self._forget_ahead(restore=True)
        return self._voices(start, n)

    def render_ahead(self, start, n) -> None:
        saved = DeviceBuffer(self.state.shape, self.state.dtype)
        last_end = self.last_end
        self.ahead = (start, n, self._voices(start, n, backup=saved), (saved, last_end))

    def sum_voices(self, voices, n):
        out = DeviceBuffer((self.k, n, self.ch), np.float32)
        check(lib().pgx_supersaw_sum(out.ptr, n * self.ch, self.k, self.nv, n, self.ch, voices.ptr,
                                     self.amp.ptr, None, 0), "pgx_supersaw_sum")
        return out

    def banked(self, n: int) -> bool:
        """The voices-summed-on-chip kernel with per-voice tables and double-buffered states (pgx_supersaw_bank_seg):
        in time segments for a few instances, one segment each from FUSED_SUPERSAW_MIN instances on."""
        return self.segmented(n) or (self.fused() and self.closed_form_ok)

    def _bank(self, start, n):
        """[instances][n][channels] on the current stream; the states move from `state` to `state_alt` and swap."""
        L = lib()
        if self.ahead is not None:
            self._forget_ahead(restore=True)
        if self.last_end is None or start != self.last_end:
            self.state.upload(self.init_state)
        if self.state_alt is None:
            self.state_alt = DeviceBuffer(self.state.shape, self.state.dtype)
        kind = "wide" if self.wide() else "bank"
        tables = self.tables.get(kind)
        if tables is None:
            size, make = ((L.pgx_supersaw_wide_table_bytes, L.pgx_supersaw_wide_tables) if kind == "wide"
                          else (L.pgx_supersaw_bank_table_bytes, L.pgx_supersaw_bank_tables))
            tables = self.tables[kind] = DeviceBuffer((size(self.k, self.nv),), np.uint8)
            check(make(tables.ptr, self.k, self.nv, self.sr, self.params.ptr), f"pgx_supersaw_{kind}_tables")
        out = DeviceBuffer((self.k, n, self.ch), np.float32)
        if kind == "wide":
            check(L.pgx_supersaw_wide(out.ptr, n * self.ch, self.k, self.nv, n, self.ch, self.state.ptr,
                                      self.state_alt.ptr, self.amp.ptr, tables.ptr), "pgx_supersaw_wide")
        else:
            check(L.pgx_supersaw_bank_seg(out.ptr, n * self.ch, self.k, self.nv, n, self.ch, self.sr,
                                          self.params.ptr, self.state.ptr, self.state_alt.ptr, self.amp.ptr,
                                          tables.ptr), "pgx_supersaw_bank_seg")
        self.state, self.state_alt = self.state_alt, self.state
        self.last_end = start + n
        return out

    def _forget_bank_ahead(self, restore: bool) -> None:
        ahead, self.ahead_bank = self.ahead_bank, None
        if ahead is not None and restore:
            # one render happened since: the states it started from are the other buffer
            self.state, self.state_alt = self.state_alt, self.state
            self.last_end = ahead[3]

    def take_bank(self, start, n):
        """The bank's output for (start, n): the block rendered ahead if it is this one, else rendered now (after the
        states went back to where the caller's last block left them)."""
        if self.ahead_bank is not None:
            if self.ahead_bank[0] == start and self.ahead_bank[1] == n:
                out, self.ahead_bank = self.ahead_bank[2], None
                return out
            self._forget_bank_ahead(restore=True)
        return self._bank(start, n)

    def render_bank_ahead(self, start, n) -> None:
        last_end = self.last_end
        self.ahead_bank = (start, n, self._bank(start, n), last_end)

    def render(self, start, n):
        L = lib()
        if self.banked(n):
            return self.take_bank(start, n)
        if self.ahead_bank is not None:
            self._forget_bank_ahead(restore=True)
        if self.fused():
            # enough instances to fill the chip with one wave per oscillator: voices summed on chip
            if self.last_end is None or start != self.last_end:
                self.state.upload(self.init_state)
            out = DeviceBuffer((self.k, n, self.ch), np.float32)
            check(L.pgx_supersaw_bank(out.ptr, n * self.ch, self.k, self.nv, n, self.ch, self.sr,
                                      self.params.ptr, self.state.ptr, self.amp.ptr), "pgx_supersaw_bank")
            self.last_end = start + n
            return out
        return self.sum_voices(self.take_voices(start, n), n)


class _BiquadNode(_Node):
    _STATE = ("state",)

    def __init__(self, pes, children):
        super().__init__(pes, children)
        coef = np.array([rbj_coefficients(pe._mode, pe._frequency, pe._q, pe._gain_db, self.sr)
                         for pe in pes], dtype=np.float64)
        self.coef = DeviceBuffer.from_host(coef)
        # one warm-up horizon for the batch (0 = some voice decays too slowly: exact path).  It only matters
        # for small banks, where the library cuts each voice into time segments to fill the machine; a
        # 512-voice bank is one workgroup per voice either way.
        settles = [settle_frames(c[3], c[4]) for c in coef]
        self.settle = 0 if min(settles) == 0 else max(settles)
        fine = [settle_frames_fine(c[3], c[4]) for c in coef] if self.settle else [0]
        self.settle_fine = 0 if min(fine) == 0 else max(fine)      # (a multiple of 16: the on-chip mix's warm-up)
        self.rot_tables = None
        self.tiles_ws = None
        self.entries_ahead = {}      # start -> (n, set): what the tiles of that block enter with, made behind the block before it
        self.tables = None           # per-voice powers of A (pgx_biquad_tables), made on first render
        self.state = None
        self.state_alt = None        # (the time-segmented chain reads one buffer and writes the other)
        self.ws = None

    def reset(self):
        super().reset()
        if self.state is not None:
            self.state.zero_()

    def channels(self):
        return self.children["source"].channels()

    def _chain_segments(self, n: int) -> int:
        """Time segments of the fused oscillator -> filter launch for a small bank, or 0: not that path."""
        src = self.children["source"]
        if not (SEGMENTED_CHAIN and isinstance(src, _BlitSawNode) and src.ch == 1 and 4 <= self.k <= SEGMENTED_CHAIN_MAX
                and src.wide() and src.closed_form_ok and self.settle > 0):
            return 0
        segs = lib().pgx_blitsaw_biquad_wide_segments(self.k, n, self.settle)
        return segs if segs > 1 else 0

    def mixes_on_chip(self, n: int) -> bool:
        """render_mix(start, n[, gain]) returns the MixPE's block: pgx_voice_tiles, no [voices][frames] layer."""
        src = self.children["source"]
        return (VOICE_TILES and isinstance(src, _BlitSawNode) and src.ch == 1 and self.k >= VOICE_TILES_MIN_VOICES and src.wide()
                and src.closed_form_ok and n >= VOICE_TILES_MIN_FRAMES
                and 0 < self.settle_fine <= VOICE_TILES_MAX_WARM)

    def quiesce(self, keep=None):
        mine = self.entries_ahead.get(keep[0]) if keep is not None else None
        self.entries_ahead = {keep[0]: mine} if mine is not None and mine[0] == keep[1] else {}

    def _mix_tables(self):
        L = lib()
        src = self.children["source"]
        saw_tables = src.tables.get("wide")
        if saw_tables is None:
            saw_tables = src.tables["wide"] = DeviceBuffer((L.pgx_supersaw_wide_table_bytes(self.k, 1),), np.uint8)
            check(L.pgx_supersaw_wide_tables(saw_tables.ptr, self.k, 1, self.sr, src.params.ptr),
                  "pgx_supersaw_wide_tables")
        if self.tables is None:
            self.tables = DeviceBuffer((self.k, L.pgx_biquad_table_doubles()), np.float64)
            check(L.pgx_biquad_tables(self.tables.ptr, self.coef.ptr, self.k), "pgx_biquad_tables")
        if self.rot_tables is None:
            assert L.pgx_voice_tiles_max_warm() == VOICE_TILES_MAX_WARM
            self.rot_tables = DeviceBuffer((L.pgx_voice_tiles_table_bytes(self.k),), np.uint8)
            check(L.pgx_voice_tiles_tables(self.rot_tables.ptr, saw_tables.ptr, self.coef.ptr, self.tables.ptr, self.k),
                  "pgx_voice_tiles_tables")

    def render_mix(self, start, n, gain=None, streaming=False):
        """The voices' mix (times `gain`, [K][n] float32, voice by voice) as one (n, 1) block.  streaming: (start + n, n)
        is expected next -- what its tiles enter with (the oscillators' phases decide it: pgx_voice_tiles_entries) is made
        in the launch that adds this block's rows, off the next block's critical chain."""
        L = lib()
        src = self.children["source"]
        if self.state is None:
            self.state = DeviceBuffer((self.k, 1, 2), np.float64, zero=True)
        if self.state_alt is None:
            self.state_alt = DeviceBuffer(self.state.shape, np.float64)
        if src.state_alt is None:
            src.state_alt = DeviceBuffer(src.state.shape, src.state.dtype)
        slot = -1
        mine = self.entries_ahead.pop(start, None)
        if mine is not None and mine[0] == n and src.last_end == start:
            slot = mine[1]
        stale = [s0 for s0, (n0, _) in self.entries_ahead.items() if s0 != start + n or n0 != n]
        if stale:
            self.entries_ahead = {}
        src.prepare(start)
        self._mix_tables()
        need = L.pgx_voice_tiles_workspace_bytes(self.k, n, self.settle_fine)
        if self.tiles_ws is None or self.tiles_ws.nbytes < need:
            self.tiles_ws = DeviceBuffer((need,), np.uint8)
            slot = -1
        out = DeviceBuffer((n, 1), np.float32)
        nxt = ((slot if slot >= 0 else 0) ^ 1) if streaming else -1
        check(L.pgx_voice_tiles(out.ptr, self.k, n, self.rot_tables.ptr, src.state.ptr, src.state_alt.ptr,
                                self.state.ptr, self.state_alt.ptr, ptr(gain), n, self.settle_fine, self.tiles_ws.ptr,
                                slot, nxt), "pgx_voice_tiles")
        if streaming:
            self.entries_ahead[start + n] = (n, nxt)
        src.state, src.state_alt = src.state_alt, src.state
        self.state, self.state_alt = self.state_alt, self.state
        src.last_end = start + n
        return out

    def takes_gain(self) -> bool:
        """render(..., gain=<[K][n] float32>) multiplies the voices by it inside the chain's kernel."""
        src = self.children["source"]
        return isinstance(src, _BlitSawNode) and src.ch == 1 and self.k >= FUSED_VOICE_MIN and src.wide()

    def _render_chain_segments(self, start, n, gain=None):
        """A small bank (a rank's share of C5): oscillator -> filter as ONE launch in concurrent time segments instead of
        the segmented oscillator bank followed by the batched settled biquad (two kernels and the dispatch gap between
        them on the block's critical chain).  States are read from one buffer and written to another."""
        L = lib()
        src = self.children["source"]
        if self.state is None:
            self.state = DeviceBuffer((self.k, 1, 2), np.float64, zero=True)
        if self.state_alt is None:
            self.state_alt = DeviceBuffer(self.state.shape, np.float64)
        if src.state_alt is None:
            src.state_alt = DeviceBuffer(src.state.shape, src.state.dtype)
        src.prepare(start)
        saw_tables = src.tables.get("wide")
        if saw_tables is None:
            saw_tables = src.tables["wide"] = DeviceBuffer((L.pgx_supersaw_wide_table_bytes(self.k, 1),), np.uint8)
            check(L.pgx_supersaw_wide_tables(saw_tables.ptr, self.k, 1, self.sr, src.params.ptr),
                  "pgx_supersaw_wide_tables")
        if self.tables is None:
            self.tables = DeviceBuffer((self.k, L.pgx_biquad_table_doubles()), np.float64)
            check(L.pgx_biquad_tables(self.tables.ptr, self.coef.ptr, self.k), "pgx_biquad_tables")
        out = DeviceBuffer((self.k, n, 1), np.float32)
        check(L.pgx_blitsaw_biquad_wide_seg(out.ptr, n, self.k, n, saw_tables.ptr, src.state.ptr, src.state_alt.ptr,
                                            self.coef.ptr, self.tables.ptr, self.state.ptr, self.state_alt.ptr,
                                            ptr(gain), n, self.settle), "pgx_blitsaw_biquad_wide_seg")
        src.state, src.state_alt = src.state_alt, src.state
        self.state, self.state_alt = self.state_alt, self.state
        src.last_end = start + n
        return out

    def render(self, start, n, gain=None):
        L = lib()
        src = self.children["source"]
        if self._chain_segments(n):
            return self._render_chain_segments(start, n, gain)
        if isinstance(src, _BlitSawNode) and src.ch == 1 and self.k >= FUSED_VOICE_MIN:
            # oscillator -> filter without the [K][frames] oscillator buffer (pgx_blitsaw_biquad_bank)
            if self.state is None:
                self.state = DeviceBuffer((self.k, 1, 2), np.float64, zero=True)
            src.prepare(start)
            out = DeviceBuffer((self.k, n, 1), np.float32)
            if src.wide():
                # sixteen frames per thread (pgx_blitsaw_biquad_wide): the oscillator's and the filter's per-voice tables
                saw_tables = src.tables.get("wide")
                if saw_tables is None:
                    saw_tables = src.tables["wide"] = DeviceBuffer((L.pgx_supersaw_wide_table_bytes(self.k, 1),), np.uint8)
                    check(L.pgx_supersaw_wide_tables(saw_tables.ptr, self.k, 1, self.sr, src.params.ptr),
                          "pgx_supersaw_wide_tables")
                if self.tables is None:
                    self.tables = DeviceBuffer((self.k, L.pgx_biquad_table_doubles()), np.float64)
                    check(L.pgx_biquad_tables(self.tables.ptr, self.coef.ptr, self.k), "pgx_biquad_tables")
                check(L.pgx_blitsaw_biquad_wide(out.ptr, n, self.k, n, saw_tables.ptr, src.state.ptr, self.coef.ptr,
                                                self.tables.ptr, self.state.ptr, ptr(gain), n),
                      "pgx_blitsaw_biquad_wide")
            else:
                check(L.pgx_blitsaw_biquad_bank(out.ptr, n, self.k, n, self.sr, src.params.ptr, src.state.ptr,
                                                self.coef.ptr, self.state.ptr), "pgx_blitsaw_biquad_bank")
            src.last_end = start + n
            return out
        x = src.render(start, n)
        ch = x.shape[2]
        if self.state is None:
            self.state = DeviceBuffer((self.k, ch, 2), np.float64, zero=True)
        if self.tables is None:
            self.tables = DeviceBuffer((self.k, L.pgx_biquad_table_doubles()), np.float64)
            check(L.pgx_biquad_tables(self.tables.ptr, self.coef.ptr, self.k), "pgx_biquad_tables")
        need = L.pgx_biquad_workspace_bytes(self.k, n, ch, self.settle)
        if need and (self.ws is None or self.ws.nbytes < need):
            self.ws = DeviceBuffer((need,), np.uint8)
        out = DeviceBuffer((self.k, n, ch), np.float32)
        check(L.pgx_biquad_const(out.ptr, n * ch, x.ptr, n * ch, self.k, n, ch, self.coef.ptr,
                                 self.tables.ptr, self.settle, self.state.ptr, ptr(self.ws) if need else None),
              "pgx_biquad_const")
        return out


class _LadderNode(_Node):
    """Bank of LadderPEs.  The ladder kernel is latency-bound (two waves per CU, most of the chip idle) and its
    input, a bank of scalar oscillators, is a pure function of time plus a few carried numbers -- so while block k
    goes through the ladder on the main stream, the oscillators of block k+1 are rendered on the side stream.
    The speculation is undone exactly if the next pull is not the next block: the oscillator states are
    snapshot before it and copied back."""

    _STATE = ("state",)

    def quiesce(self, keep=None):
        self._settle_window()
        self._forget_ahead(restore=True)

    def __init__(self, pes, children):
        super().__init__(pes, children)
        rec = np.zeros(self.k, dtype=_dev.LADDER_PARAMS)
        for i, pe in enumerate(pes):
            for key, v in pe._scalar_params().items():
                rec[i][key] = v
        self.params = _dev.upload_structs(rec)
        self.state = None
        self.ws = None
        settles = [pe._settle_frames() for pe in pes]
        self.settle = 0 if min(settles) == 0 else max(settles)     # one warm-up length for the batch
        self.accurate = max(pe._accurate_frames() for pe in pes) if self.settle else 0
        # a voice without an estimate (at or above self-oscillation, a very slow decay): warm-up lengths by trial, the
        # device check decides (ladder_pe.SettleOptimist) -- an oscillator bank locks a saturating ladder to itself
        from .ladder_pe import SettleOptimist
        self.optimist = SettleOptimist() if self.settle == 0 else None
        self.ahead = None          # (start, n, input buffer, (state copy, last_end)) rendered ahead of the caller
        self.is_root = False       # directly under the bank's mix: may hand out rows of a window (VoiceBank.__init__)
        self.win = None            # [first, end, n, buffer, served, (ladder state, oscillator state, last_end)]
        self.last = None           # (start, n) of the last block handed out
        self.grow = LADDER_WINDOW_FIRST

    # ---- windows (root node only).  A lane of k_ladder_segments integrates 1024 warm-up samples to emit its share of the
    # block -- 94 frames of a 48 000-frame block of 64 instances: 92 % of the work is warm-up.  A stream of equal blocks
    # is therefore rendered several blocks at a time (2, 4, 8: the share per lane grows, the warm-up does not) and handed
    # out block by block as rows of that window.  A pull that is not the next block puts the states back to the
    # window's start and renders the consumed part again, quietly (exact: the same kernels over the same frames).
    def _settle_window(self) -> None:
        win, self.win = self.win, None
        if win is None:
            return
        first, end, n, buf, served, (ladder_state, osc_state, osc_last_end) = win
        if served >= end:
            return                                       # consumed to the last frame: the states are already there
        self._forget_ahead(restore=False)
        src = self.children["source"]
        check(lib().pgx_memcpy_d2d(self.state.ptr, ladder_state.ptr, ladder_state.nbytes), "pgx_memcpy_d2d")
        check(lib().pgx_memcpy_d2d(src.state.ptr, osc_state.ptr, osc_state.nbytes), "pgx_memcpy_d2d")
        src.last_end = osc_last_end
        if served > first:
            self._render_now(first, served - first)

    def render(self, start, n):
        win = self.win
        if win is not None:
            if start == win[4] and n == win[2] and start + n <= win[1]:
                win[4] = start + n
                self.last = (start, n)
                return _Rows(win[3], start - win[0], n)
            self._settle_window()
        streaming = self.last == (start - n, n)
        self.last = (start, n)
        src = self.children["source"]
        if (LADDER_WINDOWS and self.is_root and streaming and n >= 4096 and self.state is not None
                and isinstance(src, (_SuperSawNode, _BlitSawNode)) and not lib().pgx_stream_is_forked()):
            blocks = max(1, min(self.grow, LADDER_WINDOW_FRAMES // n))
            if blocks > 1:
                self.grow = min(self.grow * 2, LADDER_WINDOW_MAX)
                if self.ahead is not None and not (self.ahead[0] == start and self.ahead[1] == n * blocks):
                    self._forget_ahead(restore=True)
                L = lib()
                snap = (DeviceBuffer(self.state.shape, self.state.dtype), DeviceBuffer(src.state.shape, src.state.dtype))
                check(L.pgx_memcpy_d2d(snap[0].ptr, self.state.ptr, snap[0].nbytes), "pgx_memcpy_d2d")
                if self.ahead is not None:               # the oscillators already ran on: their states before that
                    check(L.pgx_memcpy_d2d(snap[1].ptr, self.ahead[3][0].ptr, snap[1].nbytes), "pgx_memcpy_d2d")
                    osc_last_end = self.ahead[3][1]
                else:
                    check(L.pgx_memcpy_d2d(snap[1].ptr, src.state.ptr, snap[1].nbytes), "pgx_memcpy_d2d")
                    osc_last_end = src.last_end
                big = self._render_now(start, n * blocks)
                from . import look_ahead as _look_ahead
                _look_ahead.STATS["window_frames"] += n * blocks          # (bench.py: frames rendered vs frames counted)
                _look_ahead.STATS["windows"] += 1
                self.win = [start, start + n * blocks, n, big, start + n, (snap[0], snap[1], osc_last_end)]
                return _Rows(big, 0, n)
        else:
            self.grow = LADDER_WINDOW_FIRST
        return self._render_now(start, n)

    def _forget_ahead(self, restore: bool) -> None:
        ahead, self.ahead = self.ahead, None
        if ahead is not None and restore:
            src = self.children["source"]
            saved, last_end = ahead[3]
            check(lib().pgx_memcpy_d2d(src.state.ptr, saved.ptr, saved.nbytes), "pgx_memcpy_d2d")
            src.last_end = last_end

    def reset(self):
        self.win = None
        self.last = None
        self.grow = LADDER_WINDOW_FIRST
        self._forget_ahead(restore=False)            # the oscillators start over anyway
        super().reset()
        if self.state is not None:
            self.state.zero_()

    def channels(self):
        return self.children["source"].channels()

    def _render_now(self, start, n):
        L = lib()
        src = self.children["source"]
        x = None
        if self.ahead is not None:
            if self.ahead[0] == start and self.ahead[1] == n:
                x, self.ahead = self.ahead[2], None
            else:
                self._forget_ahead(restore=True)
        if x is None:
            x = src.render(start, n)
        ch = x.shape[2]
        if self.state is None:
            self.state = DeviceBuffer((self.k, ch, 9), np.float64, zero=True)
        out = DeviceBuffer((self.k, n, ch), np.float32)
        settle, accurate = self.settle, self.accurate
        if self.optimist is not None and n >= 8192:
            settle, accurate = self.optimist.settle(n)
        need = L.pgx_ladder_workspace_bytes(self.k, n, ch, settle)
        if need and (self.ws is None or self.ws.nbytes < need):
            counters = None if self.ws is None else self.ws.rows(0, 16)
            self.ws = DeviceBuffer((need,), np.uint8, zero=True)
            if counters is not None:                     # cumulative counters at the head of the workspace
                check(L.pgx_memcpy_d2d(self.ws.ptr, counters.ptr, 16), "pgx_memcpy_d2d")

        def ladder():
            check(L.pgx_ladder(out.ptr, n * ch, x.ptr, n * ch, self.k, n, ch, self.sr, self.params.ptr,
                               None, None, None, self.state.ptr, settle, accurate,
                               ptr(self.ws) if need else None), "pgx_ladder")
            if self.optimist is not None and need and settle:
                self.optimist.launched(self.ws, settle, n, self.k * ch)

        speculate = (PREFETCH_LADDER_INPUT and isinstance(src, (_SuperSawNode, _BlitSawNode)) and n >= 4096
                     and not L.pgx_stream_is_forked())
        if not speculate:
            ladder()
            return out
        # The ladder stays on the main stream, in front of everything else of this block: no cross-stream wait on
        # the chain ladder -> finish -> mix -> next ladder, and its waves are placed before the oscillators' are.
        check(L.pgx_stream_fork(), "pgx_stream_fork")              # side stream: behind x
        try:
            check(L.pgx_stream_select(0), "pgx_stream_select")
            ladder()
            check(L.pgx_stream_select(1), "pgx_stream_select")
            saved = DeviceBuffer(src.state.shape, src.state.dtype)
            check(L.pgx_memcpy_d2d(saved.ptr, src.state.ptr, saved.nbytes), "pgx_memcpy_d2d")
            snapshot = (saved, src.last_end)
            nxt = src.render(start + n, n)                         # side stream, next to the ladder
        finally:
            check(L.pgx_stream_join(), "pgx_stream_join")          # what follows the ladder also follows the oscillators
        self.ahead = (start + n, n, nxt, snapshot)
        return out


class _CombNode(_Node):
    """Bank of CombPEs with scalar frequency and feedback: every voice's D * C polyphase chains in one launch
    (two when the block is long enough to be cut into time segments), per-voice delay / feedback / ring."""

    _STATE = ("ring", "total", "parity")

    def __init__(self, pes, children):
        super().__init__(pes, children)
        rec = np.concatenate([pe._param_record() for pe in pes])
        self.params = _dev.upload_structs(rec)
        self.d_min, self.d_max = int(rec["delay"].min()), int(rec["delay"].max())
        self.rows = int(rec["buffer_len"].max())
        self.ring = None
        self.ws = None
        self.total = 0
        self.parity = 0

    def reset(self):
        super().reset()
        # CombPE has no _reset_state hook (comb_pe.py): only start / stop clear the ring -- VoiceBank.reset is
        # called from those
        self.ring = None

    def channels(self):
        return self.children["source"].channels()

    def render(self, start, n):
        L = lib()
        x = self.children["source"].render(start, n)
        ch = x.shape[2]
        if self.ring is None:
            self.ring = DeviceBuffer((self.k, 2, self.rows, ch), np.float64, zero=True)
            self.total, self.parity = 0, 0
        need = L.pgx_comb_workspace_bytes(self.k, n, ch, self.d_max, 0)
        if need and (self.ws is None or self.ws.nbytes < need):
            self.ws = DeviceBuffer((need,), np.uint8)
        out = DeviceBuffer((self.k, n, ch), np.float32)
        check(L.pgx_comb(out.ptr, n * ch, x.ptr, n * ch, self.k, n, ch, self.sr, self.params.ptr, self.d_min,
                         self.d_max, None, None, 1.0, 1, self.ring.ptr, self.rows, self.total, self.parity, None,
                         ptr(self.ws) if need else None), "pgx_comb")
        self.total += n
        self.parity ^= 1
        return out


class _GateNode(_Node):
    def __init__(self, pes):
        super().__init__(pes, {})
        rec = np.zeros(self.k, dtype=_dev.GATE_PARAMS)
        for i, pe in enumerate(pes):
            for key, v in pe._gate_params().items():
                rec[i][key] = v
        self.params = _dev.upload_structs(rec)

    def channels(self):
        return 1

    def render(self, start, n):
        out = DeviceBuffer((self.k, n, 1), np.float32)
        check(lib().pgx_periodic_gate(out.ptr, n, self.k, start, n, self.params.ptr), "pgx_periodic_gate")
        return out


class _AdsrGatedNode(_Node):
    _STATE = ("state", "last")

    def quiesce(self, keep=None):
        """keep = (start, n): the render that follows -- envelopes walked ahead for exactly that render stay (they were
        made from `state`, which they leave alone: a snapshot taken now is the state before them)."""
        if keep is not None and self.ahead is not None and self.ahead[:2] == tuple(keep):
            return
        self.forget_ahead()

    def __init__(self, pes, children):
        super().__init__(pes, children)
        rec = np.zeros(self.k, dtype=_dev.ADSR_PARAMS)
        for i, pe in enumerate(pes):
            for key, v in pe._adsr_params().items():
                rec[i][key] = v
        self.params = _dev.upload_structs(rec)
        self.state = DeviceBuffer((self.k, 3), np.float64, zero=True)
        self.ws = None
        self.ahead = None            # (start, n, envelopes): render_ahead
        self.state_next = None       # ... which reads `state` and leaves the states after the block here
        self.ring = []               # ... into one of three envelope buffers of its own, used in turn: [buffer, event]
        self.ring_at = 0             #     (event: recorded behind the mix that read the buffer last; None: never used)
        self.last = None             # (start, n) of the last block handed out

    def _scratch(self, n):
        need = lib().pgx_adsr_workspace_bytes(self.k, n)
        if self.ws is None or self.ws.nbytes < need:
            if self.ahead is not None:                          # (the side stream may still be using the old one)
                check(lib().pgx_stream_wait_detached(), "pgx_stream_wait_detached")
            self.ws = DeviceBuffer((need,), np.uint8)
        return self.ws

    def reset(self):
        self.forget_ahead()
        self.last = None
        super().reset()
        self.state.zero_()

    # ---- one block ahead (VoiceBank.render_mix): the envelopes depend on nothing but time and three carried numbers
    # each, and their walk is a latency chain that leaves the machine to everybody else -- so block k+1's are walked
    # on the side stream while block k is mixed and block k+1's oscillators run; nobody waits for them until block
    # k+1's mix.  A pull that is not the next block puts the states back.
    def forget_ahead(self, restore: bool = True) -> None:
        """(restore: nothing to do -- a block rendered ahead reads `state` and writes `state_next`)"""
        ahead, self.ahead = self.ahead, None
        if ahead is not None:
            check(lib().pgx_stream_wait_detached(), "pgx_stream_wait_detached")

    def take_ahead(self, start, n):
        """The envelopes of (start, n) if they were rendered ahead (the main stream then waits for them, which by now is
        no wait) -- their end states become the carried ones; else None, nothing having moved."""
        ahead = self.ahead
        if ahead is None:
            return None
        self.forget_ahead()
        if ahead[0] == start and ahead[1] == n:
            self.state, self.state_next = self.state_next, self.state
            self.last = (start, n)
            return ahead[2]
        return None

    def render_ahead(self, start, n, also=None, edges_here=False, behind_main=False) -> None:
        """Fused PeriodicGate only.  Edge search and walk go to the side stream, which starts behind what the main
        stream holds so far and is left running (pgx_stream_detach).  also: called on the side stream, in front of the
        walk (the on-chip mix's entries of the same block)."""
        L = lib()
        gate_node = self.children["gate"]
        if self.state_next is None:
            self.state_next = DeviceBuffer(self.state.shape, self.state.dtype)
        scratch = self._scratch(n)
        # The walk depends on nothing the main stream is doing: it only must not overwrite an envelope buffer a mix is
        # still reading.  With three buffers of the node's own used in turn, the last reader of the one to be written is
        # the mix of three blocks ago, long finished: the side stream waits for THAT (an event recorded behind it,
        # mark_consumed) instead of for the main stream's tail -- no queue to be woken (~17 us), and the walk starts as
        # soon as the one before it has ended.  A buffer's first use starts behind the main stream's tail (it comes
        # from the pool: its previous owner's kernels are somewhere on the main stream).
        # (A full bank's walk -- 512 envelopes x 8 waves -- is better started behind the block's oscillators, whose SIMDs it
        # would share from their first tile otherwise, and written to the pool's most recently freed buffer: three
        # 98 MB buffers in turn are more than the memory-side cache holds.  137 us per block against 144.)
        if self.k > EARLY_WALK_MAX_VOICES:
            out, seen = DeviceBuffer((self.k, n, 1), np.float32), None
            if edges_here and also is None:
                # (the caller enqueues the block's voices right behind this call: an edge search that starts beside them --
                # short, parallel -- finds no free SIMD for 20 us, and the walk waits for it.  It goes in front of them, on
                # the main stream; the library forks behind it)
                try:
                    check(L.pgx_adsr_gated_periodic_to(out.ptr, n, self.k, start, n, gate_node.params.ptr, self.params.ptr,
                                                       self.state.ptr, self.state_next.ptr, scratch.ptr, 1),
                          "pgx_adsr_gated_periodic_to")
                finally:
                    if L.pgx_stream_is_forked():
                        check(L.pgx_stream_detach(), "pgx_stream_detach")
                self.ahead = (start, n, out)
                return
        else:
            if not self.ring or self.ring[0][0].shape != (self.k, n, 1):
                self.ring = [[DeviceBuffer((self.k, n, 1), np.float32), None] for _ in range(3)]
            self.ring_at = (self.ring_at + 1) % 3
            out, seen = self.ring[self.ring_at]
        if seen is None or behind_main:       # (behind_main: the states this walk starts from were written on the main stream)
            check(L.pgx_stream_fork(), "pgx_stream_fork")
        else:
            check(L.pgx_stream_fork_after(seen.ptr), "pgx_stream_fork_after")
        try:
            if also is not None:                 # (first: short, and nothing behind it on this stream depends on it)
                also()
            check(L.pgx_adsr_gated_periodic_to(out.ptr, n, self.k, start, n, gate_node.params.ptr, self.params.ptr,
                                               self.state.ptr, self.state_next.ptr, scratch.ptr, 0),
                  "pgx_adsr_gated_periodic_to")
        finally:
            check(L.pgx_stream_detach(), "pgx_stream_detach")
        self.ahead = (start, n, out)

    def mark_consumed(self, env) -> None:
        """Called behind the launch that read `env` (a buffer take_ahead handed out): from here on the main stream is done
        with it."""
        for slot in self.ring:
            if slot[0] is env:
                if slot[1] is None:
                    slot[1] = _dev.Event()
                slot[1].record()

    def channels(self):
        return 1

    def fused_gate(self) -> bool:
        return isinstance(self.children["gate"], _GateNode)

    def render(self, start, n, detach=False):
        """detach=True (fused gate only): the envelope walk is left running on the side stream;
        the caller joins (pgx_stream_join) before using the result."""
        if self.ahead is not None:
            self.forget_ahead()
        self.last = (start, n)
        out = DeviceBuffer((self.k, n, 1), np.float32)
        gate_node = self.children["gate"]
        if isinstance(gate_node, _GateNode):
            # PeriodicGate feeding the envelope: evaluate the gate inside the envelope kernel
            check(lib().pgx_adsr_gated_periodic(out.ptr, n, self.k, start, n, gate_node.params.ptr,
                                                self.params.ptr, self.state.ptr, self._scratch(n).ptr,
                                                1 if detach else 0), "pgx_adsr_gated_periodic")
            return out
        gate = gate_node.render(start, n)
        check(lib().pgx_adsr_gated(out.ptr, n, gate.ptr, n, self.k, n, self.params.ptr, self.state.ptr,
                                   self._scratch(n).ptr), "pgx_adsr_gated")
        return out


class _GainNode(_Node):
    def __init__(self, pes, children):
        super().__init__(pes, children)
        self.gains = None
        if "gain" not in children:
            self.gains = [float(np.float32(pe._gain)) for pe in pes]

    def channels(self):
        return self.children["source"].channels()

    def render(self, start, n):
        L = lib()
        x = self.children["source"].render(start, n)
        ch = x.shape[2]
        out = DeviceBuffer((self.k, n, ch), np.float32)
        if self.gains is None:
            g = self.children["gain"].render(start, n)
            # elementwise over the stacked [K*n] frames: one launch for all voices
            check(L.pgx_gain_vec(out.ptr, x.ptr, g.ptr, self.k * n, ch, g.shape[2]), "pgx_gain_vec")
        elif len(set(self.gains)) == 1:
            check(L.pgx_gain_const(out.ptr, x.ptr, self.k * n * ch, self.gains[0]), "pgx_gain_const")
        else:
            for i, gval in enumerate(self.gains):
                check(L.pgx_gain_const(out.offset_ptr(i * n * ch), x.offset_ptr(i * n * ch), n * ch, gval),
                      "pgx_gain_const")
        return out


# ------------------------------------------------------------------------------------ builder
def on_chip_mix_rule(pes) -> bool:
    """Host-only (no device, no bank): would a bank of these voices -- or of any subset of at least VOICE_TILES_MIN_VOICES of
    them -- mix its voices on chip (_BiquadNode.mixes_on_chip, block length permitting)?  ShardedMixPE asks this of ALL the
    inputs of a sharded mix on every rank: the conditions are per voice, so what holds for all holds for every rank's share."""
    if not (VOICE_TILES and WIDE_SUPERSAW and pes):
        return False
    chains = []
    for pe in pes:
        if isinstance(pe, GainPE):
            if not pe._gain_is_pe or pe._gain.channel_count() != 1:
                return False
            pe = pe._source
        if not isinstance(pe, BiquadPE) or pe._freq_is_pe or pe._q_is_pe:
            return False
        saw = pe._source
        if not isinstance(saw, BlitSawPE) or saw.inputs() or saw._channels != 1:
            return False
        chains.append((saw, pe))
    sr = float(pes[0].sample_rate)
    rec = np.zeros(len(chains), dtype=_dev.BLITSAW_PARAMS)
    for i, (saw, _) in enumerate(chains):
        for key, v in saw._scalar_params().items():
            rec[i][key] = v
    if not (np.all(rec["m"] < 0.0) and np.all(rec["leak"] > 0.0) and np.all(rec["leak"] <= 0.9999)
            and np.all(rec["freq"] >= 1.0) and wide_oscillators_ok(rec, sr)):        # (_BlitSawNode.closed_form_ok, .wide_ok)
        return False
    for _, bq in chains:
        c = rbj_coefficients(bq._mode, bq._frequency, bq._q, bq._gain_db, sr)
        if not settle_frames(c[3], c[4]) or not 0 < settle_frames_fine(c[3], c[4]) <= VOICE_TILES_MAX_WARM:
            return False
    return True


def _signature(pe):
    """Structural key of a voice tree, or None when the tree cannot be batched."""
    if pe.extent() != Extent(None, None):
        return None          # MixPE's extent-skip rule is per input; banks need always-on voices
    if isinstance(pe, SinePE):
        if pe._has_pe_inputs():
            return None
        return ("sine", pe._channels)
    if isinstance(pe, BlitSawPE):
        if pe.inputs():
            return None
        return ("blitsaw", pe._channels)
    if isinstance(pe, SuperSawPE):
        if pe.inputs():
            return None
        return ("supersaw", pe._channels, len(pe._oscillators))
    if isinstance(pe, BiquadPE):
        if pe._freq_is_pe or pe._q_is_pe:
            return None
        sub = _signature(pe._source)
        return None if sub is None else ("biquad", sub)
    if isinstance(pe, LadderPE):
        if pe._freq_is_pe or pe._res_is_pe or pe._drive_is_pe:
            return None
        sub = _signature(pe._source)
        return None if sub is None else ("ladder", sub)
    if isinstance(pe, CombPE):
        if pe._freq_is_pe or pe._fb_is_pe:
            return None
        sub = _signature(pe._source)
        return None if sub is None else ("comb", sub)
    if isinstance(pe, PeriodicGate):
        return ("gate",) if pe.is_pure() else None        # PE-driven gates carry a phase: not batched
    if isinstance(pe, AdsrGatedPE):
        sub = _signature(pe._gate)
        return None if sub is None else ("adsr_gated", sub)
    if isinstance(pe, GainPE):
        sub = _signature(pe._source)
        if sub is None:
            return None
        if pe._gain_is_pe:
            g = _signature(pe._gain)
            return None if g is None else ("gain_pe", sub, g)
        return ("gain", sub)
    return None


def _collect_ids(pe, seen):
    if id(pe) in seen:
        return False
    seen.add(id(pe))
    return all(_collect_ids(c, seen) for c in pe.inputs())


def _build(pes):
    pe = pes[0]
    if isinstance(pe, SinePE):
        return _SineNode(pes)
    if isinstance(pe, BlitSawPE):
        return _BlitSawNode(pes)
    if isinstance(pe, SuperSawPE):
        return _SuperSawNode(pes)
    if isinstance(pe, BiquadPE):
        return _BiquadNode(pes, {"source": _build([p._source for p in pes])})
    if isinstance(pe, LadderPE):
        return _LadderNode(pes, {"source": _build([p._source for p in pes])})
    if isinstance(pe, CombPE):
        return _CombNode(pes, {"source": _build([p._source for p in pes])})
    if isinstance(pe, PeriodicGate):
        return _GateNode(pes)
    if isinstance(pe, AdsrGatedPE):
        return _AdsrGatedNode(pes, {"gate": _build([p._gate for p in pes])})
    if isinstance(pe, GainPE):
        children = {"source": _build([p._source for p in pes])}
        if pe._gain_is_pe:
            children["gain"] = _build([p._gain for p in pes])
        return _GainNode(pes, children)
    raise TypeError(type(pe).__name__)


class VoiceBank:
    def __init__(self, inputs):
        self.k = len(inputs)
        self.root = _build(list(inputs))
        if isinstance(self.root, _LadderNode):
            self.root.is_root = True
        self.mix_windows = False     # windows at the level of the mix whatever the root (set_mix_windows)
        self.win = None              # [first, end, n, mixed window (Snippet), served, [(node, snapshot)]]
        self.last = None             # (start, n) of the last block handed out
        self.streaming = False       # the block being rendered continues the one before it, same length
        self.grow = BANK_WINDOW_FIRST

    def reset(self) -> None:
        self.win = None
        self.last = None
        self.grow = BANK_WINDOW_FIRST
        self.root.reset()

    def set_mix_windows(self) -> None:
        """A rank's share of a sharded mix (sharding.ShardedMixPE): the windows are made at the level of the MIX, so that
        a window is one collective.  A ladder root then leaves its own windows (rows of the ladders' outputs, mixed block
        by block) alone: 8 instances 39.9 -> 41.2 us per block without the collective, but one collective per 8 blocks."""
        self.mix_windows = True
        if isinstance(self.root, _LadderNode):
            self.root.is_root = False

    # ---- windows.  A small bank's block (a rank's share of a sharded mix: 64 voices) is a handful of launches whose
    # fixed parts -- launch, per-workgroup tables and carries, the first tile's anchor sines, the gaps between dependent
    # kernels -- are as long as the work itself.  A stream of equal blocks is therefore rendered 2, 4, 8 blocks at a
    # time and handed out block by block as rows of the mixed window (look_ahead.py does the same for PE graphs; a bank
    # keeps its states in its nodes, so the snapshot is taken there).  A pull that is not the next block puts every
    # node's state back to the window's start and renders the consumed part again, quietly.
    def _nodes(self):
        found, stack = [], [self.root]
        while stack:
            node = stack.pop()
            found.append(node)
            stack.extend(node.children.values())
        return found

    def _settle_window(self) -> None:
        win, self.win = self.win, None
        if win is None:
            return
        first, end, n, big, served, snaps = win
        if served >= end:
            return                                       # consumed to the last frame: the states are already there
        for node in self._nodes():
            node.quiesce()
        for node, snap in snaps:
            node.restore(snap)
        if served > first:
            self._render_mix_now(first, served - first)

    def render_mix(self, start: int, duration: int) -> Snippet:
        win = self.win
        if win is not None:
            if start == win[4] and duration == win[2] and start + duration <= win[1]:
                win[4] = start + duration
                self.last = (start, duration)
                row = Snippet.window_rows(start, win[3].dev, start - win[0], duration)
                row._bank_window = True
                return row
            self._settle_window()
        streaming = self.streaming = self.last == (start - duration, duration)
        self.last = (start, duration)
        # (SuperSaw banks below the size that fills the chip: measured 44 -> 23 us per block for a rank's 64 instances,
        # 62 -> 42 for 128.  Not 256 and more -- rendered one block ahead already, a window ahead is too much thrown away
        # when the stream ends: 97 -> 155 us; not the C5 graph -- its envelope walk and mixes do not shrink with the
        # block, 56 -> 93 us for 64 voices.  A ladder root has windows of its own.)
        on_chip = VOICE_TILE_WINDOWS and self._mixes_on_chip(duration)
        if (BANK_WINDOWS and streaming and (self.k <= BANK_WINDOW_MAX_VOICES or on_chip) and duration >= 4096
                and (isinstance(self.root, _SuperSawNode) and not self.root.fused() or BANK_WINDOWS_ANY_ROOT
                     or self.mix_windows or on_chip)
                and not lib().pgx_stream_is_forked()):
            # (a ladder root -- a rank's share of C4 -- takes the ladder bank's longer windows: its lanes' warm-up does
            # not shrink with the share, so the fewer instances a rank owns the more of a short window is warm-up)
            ladder_root = isinstance(self.root, _LadderNode)
            blocks = max(1, min(self.grow, (LADDER_WINDOW_FRAMES if ladder_root else BANK_WINDOW_FRAMES) // duration))
            if blocks > 1:
                self.grow = min(self.grow * 2, LADDER_WINDOW_MAX if ladder_root else BANK_WINDOW_MAX)
                nodes = self._nodes()
                for node in nodes:
                    node.quiesce(keep=(start, duration * blocks))
                snaps = [(node, node.snapshot()) for node in nodes if node._STATE]
                big = self._render_mix_now(start, duration * blocks)
                from . import look_ahead as _look_ahead
                _look_ahead.STATS["window_frames"] += duration * blocks
                _look_ahead.STATS["windows"] += 1
                self.win = [start, start + duration * blocks, duration, big, start + duration, snaps]
                row = Snippet.window_rows(start, big.dev, 0, duration)
                row._bank_window = True
                return row
        elif not streaming:
            self.grow = BANK_WINDOW_FIRST
        return self._render_mix_now(start, duration)

    def _mixes_on_chip(self, n: int) -> bool:
        """The voices of this bank are added on chip (pgx_voice_tiles) for blocks of n frames."""
        root = self.root
        if isinstance(root, _GainNode) and root.gains is None:
            root = root.children["source"]
        return isinstance(root, _BiquadNode) and root.mixes_on_chip(n)

    def _supersaw_pipelined(self, start: int, n: int) -> Snippet:
        """A bank of SuperSawPEs under the mix (a rank's share of a sharded mix: 64 instances at G = 8 -- or all 512).
        The oscillators are the long pole and depend on nothing but time and two carried numbers each, so they
        own the main stream, one block ahead: block k's mix (and, on the oscillator-by-oscillator path, its voice
        sum) runs on the side stream beside block k+1's oscillators.  Nothing on the main stream ever waits for the side stream to catch up (the join at the end is
        enqueued behind the oscillators, which outlast sum + mix): a cross-stream wait that has to wake a stalled
        queue costs ~16 us on this part, a whole launch.  A pull that is not the next block copies the oscillator
        states back (take_voices).  What follows on the library stream -- the all-reduce of this block, a read-back --
        runs behind the next block's oscillators: one block of latency, no throughput."""
        root, L = self.root, lib()
        banked = root.banked(n)
        # main stream: this block's oscillators (already there when the previous call rendered them ahead)
        first = root.take_bank(start, n) if banked else root.take_voices(start, n)
        check(L.pgx_stream_fork(), "pgx_stream_fork")                  # side stream: behind this block's oscillators
        try:
            stacked = first if banked else root.sum_voices(first, n)
            ch = stacked.shape[2]
            out = DeviceBuffer((n, ch), np.float32)
            check(L.pgx_mix_batch(out.ptr, stacked.ptr, n * ch, self.k, n * ch), "pgx_mix_batch")
            check(L.pgx_stream_select(0), "pgx_stream_select")
            if banked:
                root.render_bank_ahead(start + n, n)                   # main stream
            else:
                root.render_ahead(start + n, n)
        finally:
            check(L.pgx_stream_join(), "pgx_stream_join")
        return Snippet(start, out)

    def _render_mix_now(self, start: int, duration: int) -> Snippet:
        root = self.root
        # (the voices-summed-on-chip bank + mix stay on one stream: with the bank one block ahead and the mix on the
        # side stream a rank's share went from 60.5 to 64.3 us -- the fork / join packets cost more than the 5 us mix)
        if (PREFETCH_SUPERSAW_VOICES and isinstance(root, _SuperSawNode) and duration >= 4096
                and (PIPELINE_FULL_SUPERSAW_BANK if root.fused()
                     else (PIPELINE_SUPERSAW_BANK or not root.banked(duration)))
                and not lib().pgx_stream_is_forked()):
            return self._supersaw_pipelined(start, duration)
        if isinstance(root, _BiquadNode) and root.mixes_on_chip(duration):
            return Snippet(start, root.render_mix(start, duration, streaming=self.streaming))
        if (isinstance(root, _GainNode) and root.gains is None and isinstance(root.children["source"], _BiquadNode)
                and root.children["source"].mixes_on_chip(duration) and root.children["gain"].channels() == 1):
            # GainPE(BiquadPE(BlitSawPE), gain=<PE>) voices: oscillator -> filter -> x gain -> mix in one kernel
            # (pgx_voice_tiles).  The gains come first -- the kernel reads them -- so envelopes with a fused gate are
            # walked one block ahead on the side stream, the next block's walk enqueued in front of this block's voices.
            L = lib()
            gain, source = root.children["gain"], root.children["source"]
            ahead = (isinstance(gain, _AdsrGatedNode) and gain.fused_gate() and ENVELOPE_AHEAD and duration >= 1024
                     and not L.pgx_stream_is_forked())
            g = None
            if ahead:
                streaming = gain.last == (start - duration, duration)      # equal blocks, one after the other
                if gain.ahead is not None:
                    g = gain.take_ahead(start, duration)
            walked_here = g is None
            if walked_here:
                g = gain.render(start, duration)        # a stream's first block, a seek, another kind of gain: rendered now
            if ahead and streaming:
                # (walked_here: this block's walk ran on the MAIN stream and left the states the next one starts from there --
                # the side stream, which otherwise only waits for its envelope buffer's last reader, has to start behind it)
                gain.render_ahead(start + duration, duration, edges_here=True, behind_main=walked_here)
            out = source.render_mix(start, duration, gain=g, streaming=ahead and streaming)
            if isinstance(gain, _AdsrGatedNode):
                gain.mark_consumed(g)
            return Snippet(start, out)
        if isinstance(root, _GainNode) and root.gains is None:
            # voices end in GainPE(x, gain=<PE>): fuse the per-voice multiply into the mix.  The gain
            # sub-graph (envelopes: few, latency-bound waves) and the signal sub-graph (oscillators and
            # filters: VALU-bound) share nothing, so they are enqueued on two streams and overlap.
            L = lib()
            gain = root.children["gain"]
            walk_ahead = False
            if isinstance(gain, _AdsrGatedNode) and gain.fused_gate():
                ahead_ok = ENVELOPE_AHEAD and duration >= 1024 and not L.pgx_stream_is_forked()
                streaming = gain.last == (start - duration, duration)      # equal blocks, one after the other
                source = root.children["source"]
                gained = False
                if ahead_ok and gain.ahead is not None:
                    # this block's envelopes were walked while the last block was mixed (the wait for the side stream is
                    # no wait by now).  A chain that takes the gain can multiply by them in its own kernel -- the mix then
                    # reads one [voices][frames] layer, not two (35 -> 17 us) -- but the oscillators then depend on the
                    # envelopes, the next walk has to start in front of them and shares their SIMDs from the first tile:
                    # 512 voices 139 us per block against 137, 256 voices 91 against 90.  Off.
                    if FUSE_GAIN_IN_CHAIN and isinstance(source, _BiquadNode) and source.takes_gain():
                        g = gain.take_ahead(start, duration)
                        if g is not None:
                            # (the next block's walk first: the oscillators now depend on this block's envelopes, so a
                            # walk that starts behind them would be finished only a whole walk after them)
                            if streaming:
                                gain.render_ahead(start + duration, duration)
                                streaming = False
                            x = source.render(start, duration, gain=g)
                            gained = True
                        else:                           # a seek: walked now, beside the oscillators
                            try:
                                g = gain.render(start, duration, detach=True)
                                x = source.render(start, duration)
                            finally:
                                if L.pgx_stream_is_forked():
                                    check(L.pgx_stream_join(), "pgx_stream_join")
                    else:
                        x = source.render(start, duration)
                        g = gain.take_ahead(start, duration)
                        if g is None:                   # a seek: walked now, behind the oscillators
                            g = gain.render(start, duration)
                else:
                    # edge search first (parallel, short), then the walk detached on the side stream
                    try:
                        g = gain.render(start, duration, detach=True)
                        x = root.children["source"].render(start, duration)
                    finally:
                        if L.pgx_stream_is_forked():        # the fork happens inside the detached render
                            check(L.pgx_stream_join(), "pgx_stream_join")
                walk_ahead = ahead_ok and streaming
                if gained:
                    if walk_ahead:
                        gain.render_ahead(start + duration, duration)
                    ch = x.shape[2]
                    out = DeviceBuffer((duration, ch), np.float32)
                    check(L.pgx_mix_batch(out.ptr, x.ptr, duration * ch, self.k, duration * ch), "pgx_mix_batch")
                    gain.mark_consumed(g)                   # (read by the chain kernel, which is behind us too)
                    return Snippet(start, out)
            else:
                check(L.pgx_stream_fork(), "pgx_stream_fork")
                try:
                    g = gain.render(start, duration)
                    check(L.pgx_stream_select(0), "pgx_stream_select")
                    x = root.children["source"].render(start, duration)
                finally:
                    check(L.pgx_stream_join(), "pgx_stream_join")
            if walk_ahead:
                # from a stream's second block on.  Between this block's oscillators and its mix: the side stream starts
                # behind what the main stream holds at the fork.  Behind the mix the walk -- edge search, walk, a
                # cross-stream wake-up: longer than the next block's oscillators -- became the critical path (a rank's 64
                # voices: 66 -> 89 us per block); in front of the oscillators it shares the SIMDs with them from their
                # first tile (66 -> 74 us; 512 voices 133 -> 143 us)
                gain.render_ahead(start + duration, duration)
            ch, gch = x.shape[2], g.shape[2]
            out = DeviceBuffer((duration, ch), np.float32)
            check(lib().pgx_gain_mix_batch(out.ptr, x.ptr, duration * ch, g.ptr, duration * gch, self.k,
                                           duration, ch, gch), "pgx_gain_mix_batch")
            if isinstance(gain, _AdsrGatedNode):
                gain.mark_consumed(g)
            return Snippet(start, out)
        stacked = root.render(start, duration)                   # [K][n][C], or rows of a window (_Rows)
        ch = stacked.shape[2]
        out = DeviceBuffer((duration, ch), np.float32)
        check(lib().pgx_mix_batch(out.ptr, stacked.ptr, getattr(stacked, "stride", duration * ch), self.k, duration * ch),
              "pgx_mix_batch")
        return Snippet(start, out)


def try_build_bank(inputs):
    """VoiceBank for `inputs` if they are >= MIN_VOICES identical, private, batchable trees."""
    if len(inputs) < MIN_VOICES:
        return None
    sig0 = _signature(inputs[0])
    if sig0 is None:
        return None
    seen = set()
    for pe in inputs:
        if _signature(pe) != sig0 or not _collect_ids(pe, seen):
            return None          # different structure, or a node shared between voices
    return VoiceBank(inputs)
