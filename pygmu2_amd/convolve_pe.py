"""
ConvolvePE: streaming convolution y = src * fir with a finite FIR PE
(convolve_pe.py:40-349).

The reference computes the linear convolution by float64 FFT overlap-save with an
(L-1)-sample input history.  Two device evaluations of the identical sum, same history
semantics (cleared when a render is not contiguous with the previous one):
  * filters shorter than FFT_MIN_TAPS: a dense Toeplitz x Hankel product on the MI355X f32
    matrix cores with float64 accumulation across 1024-tap slabs (pgx_convolve);
  * longer filters (up to 131 072 taps): hand-written float64 FFT overlap-save
    (pgx_convolve_fft), like the reference but resident in HBM.
`fft_size` is kept for API compatibility and validated like the reference does, but does
not influence the result (overlap-save output is independent of the FFT size).

Channel rules (convolve_pe.py:114-144,207-223): mono FIR -> applied to every source
channel; FIR channels == source channels -> per-channel; mono source + N-channel FIR ->
fan-out to N channels.
Deviation (documented in SURVEY.md section 8 a14): the reference cannot be re-started
after stop() (its tail is dropped and never re-created); here reset just clears history.
"""

from __future__ import annotations

import numpy as np

from ._kernels import DeviceBuffer, check, lib, new_output
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


# Crossover between the direct MFMA form (time ~ L) and the FFT form (time ~ flat at 31-36 us up to
# 65 536 taps), measured on MI355X for stereo 96 000-frame blocks (tools/conv_crossover.py): the direct
# form is level with the FFT form up to 1024 taps (36 us) and 3x behind from 2048 on.
FFT_MIN_TAPS = 2048


def _next_pow2(n: int) -> int:
    p = 1
    while p < n:
        p <<= 1
    return p


LONG_RENDER_HOPS = int(__import__("os").environ.get("PGX_FFT_LONG_HOPS", "3"))   # 0: never switch.  Measured (tools/
# c3_fft_sizes.py, 65 536 taps, stereo, us per call at 2^17 / 2^18 points): 196 609 frames 21.4 / 20.7, 393 218 29.5 / 26.3,
# 786 436 38.4 / 34.2, 1 440 000 58.8 / 53.8, 2 880 000 101.8 / 88.3 -- the larger transform wins from three hops on


def device_fft_size(fir_len: int, frames: int) -> int:
    """The transform size of the device's overlap-save for a render of `frames` frames: the smallest power of two >= twice
    the filter (every block's transform is half history), or -- for renders of at least LONG_RENDER_HOPS of its hops --
    twice that where the library has the geometry (2^18 points: a 65 536-tap filter then advances 196 609 frames per
    transform instead of 65 537, a quarter fewer transform points per frame).  The reference's `fft_size` is a speed
    knob of ITS block loop (convolve_pe.py:185-248) and does not change the linear convolution; neither does this."""
    base = int(lib().pgx_convolve_fft_size(int(fir_len)))
    if not base or not LONG_RENDER_HOPS:
        return base
    big = base * 2
    if big <= (1 << 18) and frames >= LONG_RENDER_HOPS * (base - fir_len + 1):
        return big
    return base


class ConvolvePE(ProcessingElement):
    _LOOK_AHEAD_SAFE = True            # look_ahead.py: overlap history carried sample-exactly
    _STATE_FIELDS = ("_hist", "_last_render_end")

    def __init__(self, src: ProcessingElement, fir: ProcessingElement, *, fft_size: int | None = None):
        self._src = src
        self._fir = fir
        self._fft_size = int(fft_size) if fft_size is not None else None
        self._fir_len: int | None = None
        self._h: DeviceBuffer | None = None          # (L, fir_ch) float32, rendered once
        self._fir_ch = 0
        self._out_ch = 0
        self._hist: DeviceBuffer | None = None       # (L-1, out_ch) float32
        self._workspace: DeviceBuffer | None = None
        self._device_fft = 0                         # transform size of the FFT path, 0 = direct MFMA form
        self._spectrum: DeviceBuffer | None = None   # filter spectrum of the FFT path
        self._last_render_end: int | None = None

    src = property(lambda self: self._src)
    fir = property(lambda self: self._fir)
    fft_size = property(lambda self: self._fft_size)

    def inputs(self) -> list[ProcessingElement]:
        return [self._src, self._fir]

    @staticmethod
    def ir_energy_norm(filter_pe: ProcessingElement) -> float:
        """sqrt(sum of squares) of a finite filter PE; 1.0 if unbounded or ~zero."""
        ext = filter_pe.extent()
        if ext.start is None or ext.end is None:
            return 1.0
        data = filter_pe.render(ext.start, ext.end - ext.start).data
        norm = np.sqrt(np.sum(data.astype(np.float64) ** 2))
        return float(norm) if norm > 1e-10 else 1.0

    def is_pure(self) -> bool:
        return False

    def channel_count(self) -> int | None:
        src_ch = self._src.channel_count()
        fir_ch = self._fir.channel_count()
        if src_ch is None and fir_ch is None:
            return None
        if src_ch is None:
            return fir_ch
        if fir_ch is None or int(fir_ch) == 1:
            return src_ch
        if int(src_ch) == 1:
            return int(fir_ch)
        return src_ch

    def _reset_state(self) -> None:
        self._last_render_end = None          # history is cleared on the next render

    _on_start = _reset_state
    _on_stop = _reset_state

    def _compute_extent(self) -> Extent:
        src_ext = self._src.extent()
        fir_ext = self._fir.extent()
        if fir_ext.start is not None and fir_ext.start != 0:
            raise ValueError(f"ConvolvePE filter extent must start at 0, got {fir_ext}")
        if fir_ext.start is None:
            raise ValueError(f"ConvolvePE filter extent must be finite and start at 0, got {fir_ext}")
        if fir_ext.end is None:
            raise ValueError(f"ConvolvePE filter extent must be finite, got {fir_ext}")
        length = int(fir_ext.end - fir_ext.start)
        if length < 1:
            return Extent(0, 0)
        if src_ext.end is None:
            return Extent(src_ext.start, None)
        return Extent(src_ext.start, int(src_ext.end + (length - 1)))

    def _prepare(self) -> None:
        if self._h is not None:
            return
        fir_ext = self._fir.extent()
        if fir_ext.start != 0 or fir_ext.end is None:
            raise ValueError(f"ConvolvePE filter must have extent Extent(0, N), got {fir_ext}")
        length = int(fir_ext.end)
        if length < 1:
            raise ValueError("ConvolvePE filter must be non-empty")
        h = self._fir.render(0, length)
        if h.duration != length:
            raise ValueError(f"ConvolvePE filter returned invalid shape {(h.duration, h.channels)}")
        src_ch = self._src.channel_count()
        if src_ch is None:
            src_ch = self._src.render(0, 1).channels
        fir_ch = h.channels
        if fir_ch == 1:
            out_ch = int(src_ch)
        elif int(src_ch) == 1 or fir_ch == int(src_ch):
            out_ch = fir_ch
        else:
            raise ValueError(f"ConvolvePE filter channels ({fir_ch}) must match src channels ({src_ch}), "
                             f"or be mono, or be multi-channel with a mono source.")
        if self._fft_size is None:
            self._fft_size = _next_pow2(max(2048, length))
        if self._fft_size < length:
            raise ValueError(f"fft_size ({self._fft_size}) must be >= filter length ({length})")
        self._fir_len, self._fir_ch, self._out_ch = length, fir_ch, out_ch
        self._h = h.dev
        self._hist = DeviceBuffer((max(length - 1, 1), out_ch), np.float32, zero=True)
        # Long filters: float64 FFT overlap-save on the device (the direct MFMA form costs 2*L flops per
        # sample; the FFT form a few hundred).  The library picks its own transform size; the reference's
        # fft_size only shapes ITS block loop and does not change the result.
        self._device_fft = 0
        self._spectra = {}                    # transform size -> filter spectrum (made on first use)
        if length >= FFT_MIN_TAPS:
            self._device_fft = int(lib().pgx_convolve_fft_size(length))
        if self._device_fft:
            self._spectrum_for(self._device_fft)

    def _spectrum_for(self, fft: int) -> DeviceBuffer:
        spec = self._spectra.get(fft)
        if spec is None:
            L = lib()
            spec = self._spectra[fft] = DeviceBuffer((L.pgx_convolve_fft_spectrum_bytes(fft, self._fir_ch),), np.uint8)
            check(L.pgx_convolve_fft_prepare(spec.ptr, self._h.ptr, self._fir_len, self._fir_ch, fft),
                  "pgx_convolve_fft_prepare")
        return spec

    def _render(self, start: int, duration: int) -> Snippet:
        self._prepare()
        length = self._fir_len
        if int(self._fft_size) - (length - 1) < 1:
            raise ValueError(f"fft_size ({self._fft_size}) too small for filter length ({length})")
        fresh = self._last_render_end is None or start != self._last_render_end
        if fresh and not self._device_fft:
            self._hist.zero_()                # (the FFT path is told instead: it never reads the old history)
        x = self._src.render(start, duration)
        src_ch = x.channels
        out_ch = src_ch if self._fir_ch == 1 else self._fir_ch
        if out_ch != self._out_ch:
            self._out_ch = out_ch
            self._hist = DeviceBuffer((max(length - 1, 1), out_ch), np.float32, zero=True)
            fresh = True
        if src_ch != 1 and src_ch != out_ch:
            raise ValueError(f"ConvolvePE src channels ({src_ch}) incompatible with output channels ({out_ch})")
        L = lib()
        fft = device_fft_size(length, duration) if self._device_fft else 0
        if self._device_fft:
            need = L.pgx_convolve_fft_workspace_bytes(duration, length, out_ch, fft)
        else:
            need = L.pgx_convolve_workspace_bytes(duration, length, out_ch)
        if self._workspace is None or self._workspace.nbytes < need:
            self._workspace = None
            self._workspace = DeviceBuffer((need,), np.uint8)
        out = new_output(duration, out_ch)
        if self._device_fft:
            check(L.pgx_convolve_fft(out.ptr, x.dev.ptr, duration, src_ch, self._spectrum_for(fft).ptr, length,
                                     self._fir_ch, out_ch, fft, self._hist.ptr,
                                     self._workspace.ptr, 1 if fresh else 0), "pgx_convolve_fft")
        else:
            check(L.pgx_convolve(out.ptr, x.dev.ptr, duration, src_ch, self._h.ptr, length, self._fir_ch,
                                 out_ch, self._hist.ptr, self._workspace.ptr), "pgx_convolve")
        self._last_render_end = start + duration
        return Snippet(start, out)

    def __repr__(self) -> str:
        return (f"ConvolvePE(src={type(self._src).__name__}, fir={type(self._fir).__name__}, "
                f"fft_size={self._fft_size})")
