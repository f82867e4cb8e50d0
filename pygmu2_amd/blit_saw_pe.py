"""
BlitSawPE: band-limited sawtooth (Stilson/Smith BLIT + leaky integrator)
(blit_saw_pe.py:67-299).

The whole chain -- phase accumulation (prefix scan), wrap, Dirichlet kernel, leaky
integrator (affine scan), x2, amplitude, float32 rounding -- is one kernel
(pgx_blitsaw), one workgroup per oscillator, float64 inside.  The reset rule for
non-contiguous renders (blit_saw_pe.py:182-185) is applied here, on the host.
"""

from __future__ import annotations

import numpy as np

from . import device as _dev
from ._kernels import DeviceBuffer, blitsaw_workspace, check, lib, new_output, ptr
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


WIDE_LONG_RENDERS = True     # long renders of a scalar-parameter oscillator: pgx_supersaw_wide in concurrent time segments
WIDE_MIN_FRAMES = 3 * 4096   # ... from three of its tiles on


def wide_oscillators_ok(rec, sr: float) -> bool:
    """pgx_supersaw_wide / pgx_blitsaw_biquad_wide render these oscillators (records of BLITSAW_PARAMS): the automatic
    (odd) M, a leak in (0, 0.9999] and f >= 1 Hz for the closed-form carries of the time segments, and a numerator
    recurrence n[j+1] = 2 cos(M pi inc) n[j] - n[j-1] that does not amplify its roundings: |sin(M pi inc)| >= 0.05
    (M pi inc is within 2 pi inc of pi/2 below Nyquist; above it M = 1 and the angle is pi inc itself)."""
    if not (np.all(rec["m"] < 0.0) and np.all(rec["leak"] > 0.0) and np.all(rec["leak"] <= 0.9999)
            and np.all(rec["freq"] >= 1.0)):
        return False
    f = np.asarray(rec["freq"], dtype=np.float64)
    mi = np.floor(sr / (2.0 * np.maximum(f, 1.0))).astype(np.int64)
    mi = np.maximum(mi - (1 - mi % 2), 1)                       # blit_saw_pe.py:166-173: the odd M at or below sr / 2f
    return bool(np.all(np.abs(np.sin(mi * np.pi * (f / sr))) >= 0.05))



def render_wide(pe, nv: int, start: int, duration: int, amplitude: float, channels: int):
    """A lone BlitSawPE / SuperSawPE with scalar parameters over a long block (a look-ahead window): the bank kernel
    with sixteen frames per thread, one instance, cut into as many time segments as fill the chip (closed-form integrator
    carries; voice_bank._SuperSawNode renders banks the same way).  `pe` carries _params (BLITSAW_PARAMS records of
    its nv oscillators), _state ([nv][2], read) and gets _state_alt (written) -- swapped afterwards.  None when the
    oscillators do not qualify (blit_saw_pe.wide_oscillators_ok)."""
    L = lib()
    sr = float(pe.sample_rate)
    wide = pe.__dict__.get("_wide")
    if wide is None:
        ok = wide_oscillators_ok(pe._wide_records(), sr)
        wide = pe.__dict__["_wide"] = {"ok": ok}
        if ok:
            wide["tables"] = DeviceBuffer((L.pgx_supersaw_wide_table_bytes(1, nv),), np.uint8)
            check(L.pgx_supersaw_wide_tables(wide["tables"].ptr, 1, nv, sr, pe._params.ptr), "pgx_supersaw_wide_tables")
            wide["amp"] = DeviceBuffer.from_host(np.array([float(amplitude)], dtype=np.float64))
    if not wide["ok"]:
        return None
    alt = pe.__dict__.get("_state_alt")
    if alt is None or alt.shape != pe._state.shape:
        alt = pe.__dict__["_state_alt"] = DeviceBuffer(pe._state.shape, pe._state.dtype)
    out = new_output(duration, channels)
    check(L.pgx_supersaw_wide(out.ptr, 0, 1, nv, duration, channels, pe._state.ptr, alt.ptr, wide["amp"].ptr,
                              wide["tables"].ptr), "pgx_supersaw_wide")
    pe._state, pe.__dict__["_state_alt"] = alt, pe._state
    return out


class BlitSawPE(ProcessingElement):
    _LOOK_AHEAD_SAFE = True            # look_ahead.py
    _STATE_FIELDS = ("_state", "_last_render_end")

    def __init__(self, frequency, amplitude=1.0, initial_phase: float = 0.0, m=None,
                 leak: float = 0.999, channels: int = 1):
        self._frequency = frequency
        self._amplitude = amplitude
        self._initial_phase = initial_phase % 1.0
        self._m = m
        self._leak = leak
        self._channels = channels
        self._params: DeviceBuffer | None = None
        self._state: DeviceBuffer | None = None      # {phase, integrator}
        self._last_render_end: int | None = None

    frequency = property(lambda self: self._frequency)
    amplitude = property(lambda self: self._amplitude)
    m = property(lambda self: self._m)
    leak = property(lambda self: self._leak)
    initial_phase = property(lambda self: self._initial_phase)

    def inputs(self) -> list[ProcessingElement]:
        return [p for p in (self._frequency, self._amplitude, self._m)
                if isinstance(p, ProcessingElement)]

    def is_pure(self) -> bool:
        return False

    def channel_count(self) -> int:
        return self._channels

    def _compute_extent(self) -> Extent:
        ext = Extent(None, None)
        for pe in self.inputs():
            ext = ext.intersection(pe.extent())
        return ext

    def _reset_state(self) -> None:
        self._last_render_end = None

    _on_start = _reset_state
    _on_stop = _reset_state

    # host-side pieces shared with the voice bank -------------------------------------------
    def _scalar_params(self) -> dict:
        def scalar(p):
            return 0.0 if isinstance(p, ProcessingElement) else float(p)
        if self._m is None:
            m = -1.0                                   # automatic
        elif isinstance(self._m, ProcessingElement):
            m = 1.0                                    # unused: a stream is supplied
        else:
            m = float(max(int(np.float64(float(self._m)).astype(np.int32)), 1))
        return dict(freq=scalar(self._frequency), amp=scalar(self._amplitude),
                    leak=float(self._leak), m=m)

    def _wide_records(self) -> np.ndarray:
        rec = np.zeros(1, dtype=_dev.BLITSAW_PARAMS)
        for key, v in self._scalar_params().items():
            rec[0][key] = v
        return rec

    def _initial_state(self) -> np.ndarray:
        ip = self._initial_phase
        ip = float(np.asarray(ip).reshape(-1)[0])      # SuperSaw passes a 1-element ndarray
        return np.array([ip, 0.0], dtype=np.float64)

    def _render(self, start: int, duration: int) -> Snippet:
        L = lib()
        if self._params is None:
            self._params = _dev.upload_struct(_dev.BLITSAW_PARAMS, **self._scalar_params())
        if self._state is None:
            self._state = DeviceBuffer((2,), np.float64)
            self._last_render_end = None
        if self._last_render_end is None or start != self._last_render_end:
            self._state.upload(self._initial_state())
        if (WIDE_LONG_RENDERS and duration >= WIDE_MIN_FRAMES and not self.inputs()):
            out = render_wide(self, 1, start, duration, 1.0, self._channels)     # float32(y * 2 amp * 1.0): the sample
            if out is not None:
                self._last_render_end = start + duration
                return Snippet(start, out)
        f_s, f_buf = self._control_stream(self._frequency, start, duration)
        a_s, a_buf = self._control_stream(self._amplitude, start, duration)
        m_buf = None
        if isinstance(self._m, ProcessingElement):
            _, m_buf = self._control_stream(self._m, start, duration)
        out = new_output(duration, self._channels)
        ws = blitsaw_workspace(self, 1, duration, bool(f_buf or a_buf or m_buf))
        check(L.pgx_blitsaw(out.ptr, 0, 1, duration, self._channels, float(self.sample_rate),
                            self._params.ptr, ptr(f_buf), 0, ptr(a_buf), 0, ptr(m_buf), 0,
                            self._state.ptr, ptr(ws), None), "pgx_blitsaw")
        self._last_render_end = start + duration
        return Snippet(start, out)

    def __repr__(self) -> str:
        def s(p):
            return type(p).__name__ if isinstance(p, ProcessingElement) else str(p)
        m = "auto" if self._m is None else s(self._m)
        return (f"BlitSawPE(frequency={s(self._frequency)}, amplitude={s(self._amplitude)}, "
                f"m={m}, leak={self._leak}, channels={self._channels})")
