"""
BiquadPE: RBJ-cookbook second-order IIR section, 8 modes (biquad_pe.py:65-474).

Constant frequency/Q: coefficients are computed once on the host with the reference's
own float64 formulas; the recurrence (scipy lfilter's DF-II-T order) runs on the device
as a parallel scan (pgx_biquad_const).  PE-driven frequency or Q: per-sample
coefficients and the direct-form-I recurrence of the reference's numba kernel, also a
scan, with time-varying 2x2 maps (pgx_biquad_varying).
State lives in HBM and is zeroed by on_start / on_stop / reset_state.
"""

from __future__ import annotations

from enum import Enum

import math

import numpy as np

from . import device as _dev
from . import diagnostics as _diag
from ._kernels import DeviceBuffer, check, lib, new_output, ptr
from .config import handle_error
from .extent import Extent
from .processing_element import ProcessingElement
from .sine_pe import SinePE
from .snippet import Snippet

FUSE_SINE_SOURCE = True        # BiquadPE(SinePE): generate the sine inside the filter kernel (pgx_biquad_sine)
SINE_FAST_RANGE = 2.0e9        # pgx_common.h kSinFastRange: beyond it the two PEs render separately


class BiquadMode(Enum):
    LOWPASS = "lowpass"
    HIGHPASS = "highpass"
    BANDPASS = "bandpass"
    NOTCH = "notch"
    ALLPASS = "allpass"
    PEAKING = "peaking"
    LOWSHELF = "lowshelf"
    HIGHSHELF = "highshelf"


_MODE_INDEX = {m: i for i, m in enumerate(BiquadMode)}


def rbj_coefficients(mode: BiquadMode, freq: float, q: float, gain_db: float, sample_rate: float):
    """(b0, b1, b2, a1, a2) normalised by a0, float64, for one (freq, q) pair.

    Host-side scalar evaluation of the cookbook formulas in the same operation order as
    the reference's vectorised `_compute_coefficients` (biquad_pe.py:217-335), including
    its clamps: freq to [1, 0.99*Nyquist], q to [0.01, 100].
    """
    f = np.atleast_1d(freq).astype(np.float64)      # 1-element arrays, as the reference's
    qq = np.atleast_1d(q).astype(np.float64)        # constant path passes (biquad_pe.py:369-372)
    nyquist = sample_rate / 2.0
    f = np.clip(f, 1.0, nyquist * 0.99)
    qq = np.clip(qq, 0.01, 100.0)
    omega = 2.0 * np.pi * f / sample_rate
    sn, cs = np.sin(omega), np.cos(omega)
    alpha = sn / (2.0 * qq)
    A = 10.0 ** (gain_db / 40.0)
    if mode == BiquadMode.LOWPASS:
        b0 = (1.0 - cs) / 2.0; b1 = 1.0 - cs; b2 = (1.0 - cs) / 2.0
        a0 = 1.0 + alpha; a1 = -2.0 * cs; a2 = 1.0 - alpha
    elif mode == BiquadMode.HIGHPASS:
        b0 = (1.0 + cs) / 2.0; b1 = -(1.0 + cs); b2 = (1.0 + cs) / 2.0
        a0 = 1.0 + alpha; a1 = -2.0 * cs; a2 = 1.0 - alpha
    elif mode == BiquadMode.BANDPASS:
        b0 = alpha; b1 = 0.0; b2 = -alpha
        a0 = 1.0 + alpha; a1 = -2.0 * cs; a2 = 1.0 - alpha
    elif mode == BiquadMode.NOTCH:
        b0 = 1.0; b1 = -2.0 * cs; b2 = 1.0
        a0 = 1.0 + alpha; a1 = -2.0 * cs; a2 = 1.0 - alpha
    elif mode == BiquadMode.ALLPASS:
        b0 = 1.0 - alpha; b1 = -2.0 * cs; b2 = 1.0 + alpha
        a0 = 1.0 + alpha; a1 = -2.0 * cs; a2 = 1.0 - alpha
    elif mode == BiquadMode.PEAKING:
        b0 = 1.0 + alpha * A; b1 = -2.0 * cs; b2 = 1.0 - alpha * A
        a0 = 1.0 + alpha / A; a1 = -2.0 * cs; a2 = 1.0 - alpha / A
    elif mode == BiquadMode.LOWSHELF:
        sA = np.sqrt(A)
        b0 = A * ((A + 1.0) - (A - 1.0) * cs + 2.0 * sA * alpha)
        b1 = 2.0 * A * ((A - 1.0) - (A + 1.0) * cs)
        b2 = A * ((A + 1.0) - (A - 1.0) * cs - 2.0 * sA * alpha)
        a0 = (A + 1.0) + (A - 1.0) * cs + 2.0 * sA * alpha
        a1 = -2.0 * ((A - 1.0) + (A + 1.0) * cs)
        a2 = (A + 1.0) + (A - 1.0) * cs - 2.0 * sA * alpha
    elif mode == BiquadMode.HIGHSHELF:
        sA = np.sqrt(A)
        b0 = A * ((A + 1.0) + (A - 1.0) * cs + 2.0 * sA * alpha)
        b1 = -2.0 * A * ((A - 1.0) + (A + 1.0) * cs)
        b2 = A * ((A + 1.0) + (A - 1.0) * cs - 2.0 * sA * alpha)
        a0 = (A + 1.0) - (A - 1.0) * cs + 2.0 * sA * alpha
        a1 = 2.0 * ((A - 1.0) - (A + 1.0) * cs)
        a2 = (A + 1.0) - (A - 1.0) * cs - 2.0 * sA * alpha
    else:
        handle_error(f"Unknown filter mode: {mode}", fatal=True, exception_class=ValueError)
        return (0.0, 0.0, 0.0, 0.0, 0.0)
    return tuple(float(np.atleast_1d(v)[0]) for v in (b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0))


def settle_frames(a1: float, a2: float, limit: int = 1 << 16) -> int:
    """
    Smallest power of two W <= limit for which every entry of A^W, A = [[-a1, 1], [-a2, 0]] (the
    state matrix of the DF-II-transposed section), is below 2^-90; 0 when there is none (slowly
    decaying or unstable section).  pgx_biquad_const uses it to render long blocks in one launch.
    """
    p = np.array([[-a1, 1.0], [-a2, 0.0]], dtype=np.float64)
    w = 1
    with np.errstate(over="ignore", invalid="ignore"):
        while w <= limit:
            if w >= 16 and np.all(np.abs(p) < 2.0 ** -90):
                return w
            p = p @ p
            if not np.all(np.isfinite(p)):
                return 0
            w *= 2
    return 0


def settle_frames_fine(a1: float, a2: float, step: int = 16, limit: int = 2048) -> int:
    """
    Smallest multiple of `step` W <= limit for which every entry of A^W (settle_frames' matrix) is below 2^-90; 0 when
    there is none.  settle_frames rounds up to a power of two (its callers warm up by whole tiles); the on-chip mix
    (pgx_voice_tiles) warms a filter up INSIDE a 4096-frame tile, where every 16 frames saved are frames emitted.
    """
    a = np.array([[-a1, 1.0], [-a2, 0.0]], dtype=np.float64)
    with np.errstate(over="ignore", invalid="ignore"):
        ps = np.linalg.matrix_power(a, step)
        p = ps.copy()
        w = step
        while w <= limit:
            if not np.all(np.isfinite(p)):
                return 0
            if np.all(np.abs(p) < 2.0 ** -90):
                return w
            p = p @ ps
            w += step
    return 0


def _passes_more_signal_than_rounding(coef, omega: float, horizon: int) -> bool:
    """The fused chain's sine samples are the separate SinePE's to within the rounding noise of the reference's own
    phase, but not always the same float32: a sample that rounds the other way is an impulse of one float32 ulp
    (6e-8 of the amplitude) into the filter.  The filter answers it with at most 6e-8 * sum|h| while the tone comes
    out with gain |H(e^jw)|: fusing is allowed when that leaves the error below ~1e-6 of the output's peak (a
    high-pass far above the tone, which passes the rounding noise and little else, keeps the two-launch path whose
    float32 sine is the reference's)."""
    b0, b1, b2, a1, a2 = coef
    z1 = np.exp(-1j * omega)
    gain = abs((b0 + b1 * z1 + b2 * z1 * z1) / (1.0 + a1 * z1 + a2 * z1 * z1))
    s0 = s1 = 0.0                                    # DF-II-T impulse response, sum of magnitudes
    l1, x = 0.0, 1.0
    for _ in range(max(16, min(int(horizon), 8192))):
        y = s0 + b0 * x
        s0, s1 = s1 + b1 * x - a1 * y, b2 * x - a2 * y
        l1 += abs(y)
        x = 0.0
    return bool(np.isfinite(gain) and np.isfinite(l1) and gain >= 0.05 * l1)


class BiquadPE(ProcessingElement):
    _PASSES_BLOCKS = True              # look_ahead.py: inputs are pulled with the caller's (duration)
    _LOOK_AHEAD_SAFE = True            # look_ahead.py: block-partition invariant, state listed below
    _STATE_FIELDS = ("_state", "_state_channels")

    def __init__(self, source: ProcessingElement, frequency, q,
                 mode: BiquadMode = BiquadMode.LOWPASS, gain_db: float = 0.0):
        self._source = source
        self._frequency = frequency
        self._q = q
        self._mode = mode
        self._gain_db = gain_db
        self._freq_is_pe = isinstance(frequency, ProcessingElement)
        self._q_is_pe = isinstance(q, ProcessingElement)
        self._coef: DeviceBuffer | None = None        # [5] float64 (constant path)
        self._coef_host = None
        self._sine_chain_memo = None                  # _sine_chain(), evaluated once
        self._sine_supported: dict = {}               # duration -> pgx_biquad_sine_supported
        self._backup_target = None                    # a snapshot buffer the next fused render has to fill
        self._backup_ring = []                        # three small buffers used in turn as snapshot targets
        self._backup_turn = 0
        self._settle = 0                              # settle_frames of the constant section
        self._tables: DeviceBuffer | None = None      # its power tables (single-launch path)
        self._params: DeviceBuffer | None = None      # pgx_biquad_var_params (varying path)
        self._state: DeviceBuffer | None = None       # [C][2] or [C][4] float64
        self._state_channels = 0
        self._workspace: DeviceBuffer | None = None
        self._ws_key: tuple[int, int] | None = None
        self._ws_need = 0

    source = property(lambda self: self._source)
    frequency = property(lambda self: self._frequency)
    q = property(lambda self: self._q)
    mode = property(lambda self: self._mode)
    gain_db = property(lambda self: self._gain_db)

    def inputs(self) -> list[ProcessingElement]:
        out = [self._source]
        if self._freq_is_pe:
            out.append(self._frequency)
        if self._q_is_pe:
            out.append(self._q)
        return out

    def is_pure(self) -> bool:
        return False

    def channel_count(self) -> int | None:
        return self._source.channel_count()

    def _compute_extent(self) -> Extent:
        ext = self._source.extent()
        if self._freq_is_pe:
            ext = ext.intersection(self._frequency.extent()) or ext
        if self._q_is_pe:
            ext = ext.intersection(self._q.extent()) or ext
        return ext

    def _reset_state(self) -> None:
        if self._state is not None:
            self._state.zero_()

    _on_start = _reset_state
    _on_stop = _reset_state

    def _ensure_state(self, channels: int) -> None:
        if self._state is None or self._state_channels != channels:
            per = 4 if (self._freq_is_pe or self._q_is_pe) else 2
            self._state = DeviceBuffer((channels, per), np.float64, zero=True)
            self._state_channels = channels

    def _prepare_constant(self, L, sr: float) -> None:
        if self._coef is None:
            coef = rbj_coefficients(self._mode, self._frequency, self._q, self._gain_db, sr)
            self._coef = DeviceBuffer.from_host(np.asarray(coef, dtype=np.float64))
            self._coef_host = tuple(float(c) for c in coef)
            self._settle = settle_frames(coef[3], coef[4])
            if self._settle:
                self._tables = DeviceBuffer((L.pgx_biquad_table_doubles(),), np.float64)
                check(L.pgx_biquad_tables(self._tables.ptr, self._coef.ptr, 1), "pgx_biquad_tables")

    def _sine_chain(self):
        """(w, amplitude, phase) of the source when this is BiquadPE(SinePE) with scalar parameters, mono, and the
        filter passes more signal than rounding noise -- evaluated once; None otherwise."""
        if self._sine_chain_memo is None:
            src, chain = self._source, False
            if (not self._freq_is_pe and not self._q_is_pe and type(src) is SinePE and not src._has_pe_inputs()
                    and src._channels == 1):
                L = lib()
                sr = float(self.sample_rate)
                self._prepare_constant(L, sr)
                w = 2.0 * np.pi * float(src._frequency)              # sine_pe.py: (2 pi) f, left to right
                if self._settle and _passes_more_signal_than_rounding(self._coef_host, w / sr, self._settle):
                    chain = (w, float(src._amplitude), float(src._phase))
            self._sine_chain_memo = chain
        return self._sine_chain_memo or None

    def _render_sine_source(self, start: int, duration: int):
        """BiquadPE(SinePE) with scalar parameters, mono, long block: the sine is generated inside the filter kernel
        (pgx_biquad_sine): one launch, 4 B per frame.  None when the chain does not qualify."""
        if not FUSE_SINE_SOURCE or _diag.is_enabled():
            return None
        chain = self._sine_chain()
        if chain is None:
            return None
        w, amp, phase = chain
        L = lib()
        sr = float(self.sample_rate)
        ok = self._sine_supported.get(duration)
        if ok is None:
            ok = self._sine_supported[duration] = bool(L.pgx_biquad_sine_supported(duration, self._settle))
        if not ok or abs(phase) + abs(w) * ((abs(start) + duration) / sr) >= SINE_FAST_RANGE:
            return None
        self._ensure_state(1)
        out = new_output(duration, 1)
        backup, self._backup_target = self._backup_target, None      # a look-ahead window's snapshot: the kernel fills it
        check(L.pgx_biquad_sine(out.ptr, start, duration, sr, w, amp, phase, self._coef.ptr, self._tables.ptr,
                                self._settle, self._state.ptr, ptr(backup)), "pgx_biquad_sine")
        return Snippet(start, out)

    def _la_take_snapshot(self):
        """look_ahead.take_snapshot: the state copy of a window over the fused sine chain is written by the window's
        own kernel (pgx_biquad_sine(..., state_backup)) instead of a copy launch in front of it.  Any other render
        that comes first fills the copy the ordinary way (_flush_backup)."""
        if self._state is None or self._sine_chain() is None or not FUSE_SINE_SOURCE:
            return None
        # Three buffers in turn instead of an allocation per window (1.4 us of every opening): a snapshot is used at
        # most once -- restored, it BECOMES the state -- and only the newest window's can still be restored; with the
        # live state that makes two buffers busy, the third is free.
        ring = self._backup_ring
        if len(ring) < 3 or ring[0].shape != self._state.shape:
            ring[:] = [DeviceBuffer(self._state.shape, self._state.dtype) for _ in range(3)]
        for _ in range(3):
            self._backup_turn = (self._backup_turn + 1) % 3
            target = ring[self._backup_turn]
            if target is not self._state:
                break
        self._backup_target = target
        return {"_state": target, "_state_channels": self._state_channels}

    def _flush_backup(self) -> None:
        backup, self._backup_target = self._backup_target, None
        if backup is not None and self._state is not None:
            check(lib().pgx_memcpy_d2d(backup.ptr, self._state.ptr, backup.nbytes), "pgx_memcpy_d2d")

    def _render(self, start: int, duration: int) -> Snippet:
        fused = self._render_sine_source(start, duration)
        if fused is not None:
            return fused
        if self._backup_target is not None:
            self._flush_backup()
        src = self._source.render(start, duration)
        ch = src.channels
        self._ensure_state(ch)
        out = new_output(duration, ch)
        L = lib()
        sr = float(self.sample_rate)
        if not self._freq_is_pe and not self._q_is_pe:
            self._prepare_constant(L, sr)
            if self._ws_key != (duration, ch):
                self._ws_need = L.pgx_biquad_workspace_bytes(1, duration, ch, self._settle)
                self._ws_key = (duration, ch)
            need = self._ws_need
            if need and (self._workspace is None or self._workspace.nbytes < need):
                self._workspace = DeviceBuffer((need,), np.uint8)
            check(L.pgx_biquad_const(out.ptr, 0, src.dev.ptr, 0, 1, duration, ch, self._coef.ptr,
                                     ptr(self._tables), self._settle, self._state.ptr, ptr(self._workspace) if need else None),
                  "pgx_biquad_const")
            return Snippet(start, out)

        f_s, f_buf = self._control_stream(self._frequency, start, duration)
        q_s, q_buf = self._control_stream(self._q, start, duration)
        if self._params is None:
            self._params = _dev.upload_struct(
                _dev.BIQUAD_VAR_PARAMS, freq=0.0 if f_s is None else f_s, q=0.0 if q_s is None else q_s,
                gain_db=float(self._gain_db), mode=_MODE_INDEX[self._mode])
        gain_a = 10.0 ** (self._gain_db / 40.0)
        if self._ws_key != (duration, ch):                   # (the plan only depends on the block shape)
            self._ws_need = L.pgx_scan2_workspace_bytes(duration, ch)
            self._ws_key = (duration, ch)
        need = self._ws_need
        if need and (self._workspace is None or self._workspace.nbytes < need):
            self._workspace = DeviceBuffer((need,), np.uint8)
        check(L.pgx_biquad_varying(out.ptr, src.dev.ptr, duration, ch, sr, self._params.ptr,
                                   ptr(f_buf), ptr(q_buf), gain_a, math.sqrt(gain_a),
                                   self._state.ptr, ptr(self._workspace) if need else None),
              "pgx_biquad_varying")
        return Snippet(start, out)

    def __repr__(self) -> str:
        f = f"{type(self._frequency).__name__}(...)" if self._freq_is_pe else str(self._frequency)
        q = f"{type(self._q).__name__}(...)" if self._q_is_pe else str(self._q)
        return (f"BiquadPE(source={type(self._source).__name__}, frequency={f}, q={q}, "
                f"mode={self._mode.value})")
