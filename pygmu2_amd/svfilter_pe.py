"""
SVFilterPE: trapezoidal-integration state variable filter with BiquadPE's interface
(svfilter_pe.py:290-500).

    out_n = C . [x_n, s0, s1]        s_{n+1} = B x_n + A s_n

Constant frequency/Q: (A, B, C) are evaluated once on the host in float64 with the
formulas of the reference's coefficient routine (svfilter_pe.py:106-205); PE-driven
frequency or Q: the same formulas per sample on the device.  Either way the recurrence
runs as a time-parallel 2x2 affine scan, one workgroup per channel (pgx_svf), and the
state {s0, s1} per channel lives in HBM.
"""

from __future__ import annotations

import math

import numpy as np

from . import device as _dev
from ._kernels import DeviceBuffer, check, lib, new_output, ptr
from .biquad_pe import BiquadMode
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet

_SVF_MODE_INDEX = {
    BiquadMode.LOWPASS: 0, BiquadMode.HIGHPASS: 1, BiquadMode.BANDPASS: 2, BiquadMode.NOTCH: 3,
    BiquadMode.PEAKING: 4, BiquadMode.LOWSHELF: 5, BiquadMode.HIGHSHELF: 6,
}


def svf_coefficients(mode: BiquadMode, freq: float, q: float, gain_db: float, sample_rate: float):
    """
    (a00, a01, a10, a11, b0, b1, c0, c1, c2) in float64 for one (freq, q) pair, clamped like
    the reference: freq/sr to [1e-6, 0.5], q to [0.01, 100], resonance to [0, 0.999]
    (svfilter_pe.py:120-205).
    """
    m = _SVF_MODE_INDEX[mode]
    f_norm = min(max(float(freq) / float(sample_rate), 1e-6), 0.5)
    a_lin = 10.0 ** (gain_db / 40.0)
    qc = min(max(float(q), 0.01), 100.0)
    res = 1.0 - 0.5 * (1.0 / (qc * a_lin)) if m == 4 else 1.0 - 0.5 / qc
    res = min(max(res, 0.0), 0.999)
    k = 2.0 - 2.0 * res
    g = math.tan(math.pi * f_norm)
    if m == 5:
        g = g * (1.0 / math.sqrt(a_lin))
    elif m == 6:
        g = g * math.sqrt(a_lin)
    else:
        g = g * 1.0
    a1 = 1.0 / (1.0 + g * (g + k))
    a2 = g * a1
    a3 = g * a2
    if m == 0:
        m0, m1, m2 = 0.0, 0.0, 1.0
    elif m == 1:
        m0, m1, m2 = 1.0, -k, -1.0
    elif m == 2:
        m0, m1, m2 = 0.0, 1.0, 0.0
    elif m == 3:
        m0, m1, m2 = 1.0, -k, 0.0
    elif m == 4:
        m0, m1, m2 = 1.0, k * (a_lin * a_lin - 1.0), 0.0
    elif m == 5:
        m0, m1, m2 = 1.0, k * (a_lin - 1.0), a_lin * a_lin - 1.0
    else:
        a_sq = a_lin * a_lin
        m0, m1, m2 = a_sq, k * (a_lin - a_sq), 1.0 - a_sq
    return (2.0 * a1 - 1.0, -2.0 * a2, 2.0 * a2, 1.0 - 2.0 * a3, 2.0 * a2, 2.0 * a3,
            m0 * 1.0 + m1 * a2 + m2 * a3, m1 * a1 + m2 * a2, -m1 * a2 + m2 * (1.0 - a3))


class SVFilterPE(ProcessingElement):
    _PASSES_BLOCKS = True              # look_ahead.py: inputs are pulled with the caller's (duration)
    _LOOK_AHEAD_SAFE = True            # look_ahead.py
    _STATE_FIELDS = ("_state", "_state_channels")

    def __init__(self, source: ProcessingElement, frequency, q,
                 mode: BiquadMode = BiquadMode.LOWPASS, gain_db: float = 0.0):
        if mode == BiquadMode.ALLPASS:
            raise ValueError("SVFilterPE does not support ALLPASS mode. "
                             "Use BiquadPE for allpass, or another mode.")
        self._source = source
        self._frequency = frequency
        self._q = q
        self._mode = mode
        self._gain_db = gain_db
        self._freq_is_pe = isinstance(frequency, ProcessingElement)
        self._q_is_pe = isinstance(q, ProcessingElement)
        self._coef: DeviceBuffer | None = None        # [9] float64 (constant path)
        self._params: DeviceBuffer | None = None      # pgx_biquad_var_params
        self._state: DeviceBuffer | None = None       # [C][2] float64
        self._state_channels = 0
        self._workspace: DeviceBuffer | None = None

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

    def _render(self, start: int, duration: int) -> Snippet:
        src = self._source.render(start, duration)
        ch = src.channels
        if self._state is None or self._state_channels != ch:
            self._state = DeviceBuffer((ch, 2), np.float64, zero=True)
            self._state_channels = ch
        out = new_output(duration, ch)
        sr = float(self.sample_rate)
        f_s, f_buf = self._control_stream(self._frequency, start, duration)
        q_s, q_buf = self._control_stream(self._q, start, duration)
        if self._params is None:
            self._params = _dev.upload_struct(
                _dev.BIQUAD_VAR_PARAMS, freq=0.0 if f_s is None else f_s, q=0.0 if q_s is None else q_s,
                gain_db=float(self._gain_db), mode=_SVF_MODE_INDEX[self._mode])
        constant = not self._freq_is_pe and not self._q_is_pe
        if constant and self._coef is None:
            self._coef = DeviceBuffer.from_host(np.asarray(
                svf_coefficients(self._mode, self._frequency, self._q, self._gain_db, sr), dtype=np.float64))
        L = lib()
        need = L.pgx_scan2_workspace_bytes(duration, ch)
        if need and (self._workspace is None or self._workspace.nbytes < need):
            self._workspace = DeviceBuffer((need,), np.uint8)
        check(L.pgx_svf(out.ptr, src.dev.ptr, duration, ch, sr, self._params.ptr, ptr(f_buf), ptr(q_buf),
                        10.0 ** (self._gain_db / 40.0), self._coef.ptr if constant else None,
                        self._state.ptr, ptr(self._workspace) if need else None), "pgx_svf")
        return Snippet(start, out)

    def __repr__(self) -> str:
        f = f"{type(self._frequency).__name__}(...)" if self._freq_is_pe else str(self._frequency)
        q = f"{type(self._q).__name__}(...)" if self._q_is_pe else str(self._q)
        return (f"SVFilterPE(source={type(self._source).__name__}, frequency={f}, q={q}, "
                f"mode={self._mode.value})")
