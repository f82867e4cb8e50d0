"""
SpatialPE: M -> N channel conversion and stereo / binaural placement (spatial_pe.py:34-673).

Methods, as in the reference: SpatialAdapter (channel up/down-mix), SpatialLinear and
SpatialConstantPower (stereo panning, scalar or PE-driven azimuth), SpatialHRTF (binaural, MIT
KEMAR compact set).  The first three are one element-wise kernel each.  SpatialHRTF is the mono
mix convolved with a 128-tap stereo impulse response: a direct caller of the ConvolvePE kernel
(pgx_convolve, the MFMA direct form) with the same (L-1)-sample history semantics the reference
implements with its tail.

The KEMAR recordings are data, not code, and are not shipped here: SpatialHRTF reads
`<kemar_dir>/<name>.wav` where kemar_dir is the constructor argument, else $PYGMU_KEMAR_DIR, else
pygmu2_amd/assets/kemar.  File names follow the set's published grid (Gardner & Martin 1994).
"""

from __future__ import annotations

import os
from abc import ABC, abstractmethod

import numpy as np

from ._kernels import DeviceBuffer, check, lib, new_output
from .config import handle_error
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet
from .wav_io import read_frames, read_info


class SpatialMethod(ABC):
    @property
    @abstractmethod
    def output_channels(self) -> int:
        ...

    @abstractmethod
    def render(self, source_snippet: Snippet, start: int, duration: int, sample_rate: int) -> DeviceBuffer:
        ...

    def inputs(self) -> list[ProcessingElement]:
        return []


class SpatialAdapter(SpatialMethod):
    def __init__(self, channels: int):
        if channels < 1:
            raise ValueError(f"SpatialAdapter: channels must be >= 1 (got {channels})")
        self._channels = int(channels)

    @property
    def output_channels(self) -> int:
        return self._channels

    def render(self, source_snippet, start, duration, sample_rate):
        src = source_snippet.dev
        if source_snippet.channels == self._channels:
            return src
        out = new_output(duration, self._channels)
        check(lib().pgx_channel_adapt(out.ptr, src.ptr, duration, source_snippet.channels, self._channels),
              "pgx_channel_adapt")
        return out

    def __repr__(self) -> str:
        return f"SpatialAdapter(channels={self._channels})"


class _Pan(SpatialMethod):
    _constant_power = 0

    def __init__(self, azimuth):
        self.azimuth = azimuth

    @property
    def output_channels(self) -> int:
        return 2

    def inputs(self) -> list[ProcessingElement]:
        return [self.azimuth] if isinstance(self.azimuth, ProcessingElement) else []

    def render(self, source_snippet, start, duration, sample_rate):
        out = new_output(duration, 2)
        stream = None
        if isinstance(self.azimuth, ProcessingElement):
            az = self.azimuth.render(start, duration)
            stream = az.dev
            if az.channels != 1:                       # the reference takes channel 0
                mono = DeviceBuffer((duration, 1), np.float32)
                check(lib().pgx_extract_channel(mono.ptr, stream.ptr, duration, az.channels, 0),
                      "pgx_extract_channel")
                stream = mono
        check(lib().pgx_pan(out.ptr, source_snippet.dev.ptr, duration, source_snippet.channels,
                            0.0 if stream is not None else float(self.azimuth),
                            None if stream is None else stream.ptr, self._constant_power), "pgx_pan")
        return out

    def __repr__(self) -> str:
        az = f"{self.azimuth:.1f}" if isinstance(self.azimuth, (int, float)) else type(self.azimuth).__name__
        return f"{type(self).__name__}(azimuth={az})"


class SpatialLinear(_Pan):
    _constant_power = 0


class SpatialConstantPower(_Pan):
    _constant_power = 1


# MIT KEMAR compact set: elevations -40..90 in steps of 10, this many azimuths around the full circle
_KEMAR_AZIMUTH_COUNTS = (56, 60, 72, 72, 72, 72, 72, 60, 56, 45, 36, 24, 12, 1)


def kemar_entries():
    """(elevation, azimuth, file name) of every file of the set (azimuths 0..180; the left hemisphere is the
    mirrored right one), in the order of the reference's table (spatial_pe.py:316-393)."""
    out = []
    for i, count in enumerate(_KEMAR_AZIMUTH_COUNTS):
        elev = -40 + 10 * i
        for k in range(count):
            az = int(round(k * 360.0 / count))
            if az > 180:
                break
            out.append((elev, az, f"H{elev}e{az:03d}a.wav"))
    return out


class SpatialHRTF(SpatialMethod):
    KEMAR_HRTF_ENTRIES = tuple(kemar_entries())

    @staticmethod
    def hrtf_filename_for(azimuth: float, elevation: float) -> str:
        az = min(180.0, abs(float(azimuth)))
        elev = float(elevation)
        return min(SpatialHRTF.KEMAR_HRTF_ENTRIES, key=lambda e: (e[0] - elev) ** 2 + (e[1] - az) ** 2)[2]

    def __init__(self, azimuth, elevation=0.0, *, kemar_dir: str | None = None):
        if isinstance(azimuth, ProcessingElement) or isinstance(elevation, ProcessingElement):
            raise ValueError("SpatialHRTF: azimuth and elevation must be static (float or int). "
                             "Dynamic values would switch impulse responses during rendering and cause "
                             "discontinuities.")
        self.azimuth = float(azimuth)
        self.elevation = float(elevation)
        self._kemar_dir = kemar_dir
        self._ir: DeviceBuffer | None = None         # (taps, 2) float32, columns swapped for the left side
        self._ir_len = 0
        self._ir_sr = 0
        self._hist: DeviceBuffer | None = None       # (taps - 1, 2): the reference's tail, fanned out
        self._mono: DeviceBuffer | None = None
        self._workspace: DeviceBuffer | None = None
        self._last_render_end: int | None = None
        self._warned_sr_mismatch = False

    @property
    def output_channels(self) -> int:
        return 2

    def _directory(self) -> str:
        return (self._kemar_dir or os.environ.get("PYGMU_KEMAR_DIR")
                or os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "kemar"))

    def _load_ir(self) -> None:
        if self._ir is not None:
            return
        path = os.path.join(self._directory(), self.hrtf_filename_for(self.azimuth, self.elevation))
        if not os.path.exists(path):
            raise FileNotFoundError(f"SpatialHRTF: {path} not found; pass kemar_dir= or set PYGMU_KEMAR_DIR to "
                                    "the directory holding the MIT KEMAR compact HRTF WAV files")
        info = read_info(path)
        raw = read_frames(path, info, 0, info.frames)
        if info.channels != 2:
            raise ValueError(f"SpatialHRTF: expected stereo IR, got shape {raw.shape} for {path}")
        ir = raw.astype(np.float32) if info.format_tag == 3 else raw.astype(np.float32) * np.float32(1.0 / 32768.0)
        if self.azimuth < 0:
            ir = ir[:, ::-1]
        self._ir = DeviceBuffer.from_host(np.ascontiguousarray(ir, dtype=np.float32))
        self._ir_len, self._ir_sr = info.frames, info.sample_rate
        self._hist = DeviceBuffer((max(self._ir_len - 1, 1), 2), np.float32, zero=True)

    def render(self, source_snippet, start, duration, sample_rate):
        self._load_ir()
        if sample_rate != self._ir_sr and not self._warned_sr_mismatch:
            handle_error(f"SpatialHRTF: IR sample rate is {self._ir_sr} Hz but source is {sample_rate} Hz. "
                         "Proceeding without resampling.", fatal=False)
            self._warned_sr_mismatch = True
        L = lib()
        if self._last_render_end is None or start != self._last_render_end:
            self._hist.zero_()
        mono = source_snippet.dev
        if source_snippet.channels != 1:
            mono = DeviceBuffer((duration, 1), np.float32)
            check(L.pgx_mono_mean(mono.ptr, source_snippet.dev.ptr, duration, source_snippet.channels),
                  "pgx_mono_mean")
        need = L.pgx_convolve_workspace_bytes(duration, self._ir_len, 2)
        if self._workspace is None or self._workspace.nbytes < need:
            self._workspace = DeviceBuffer((need,), np.uint8)
        out = new_output(duration, 2)
        check(L.pgx_convolve(out.ptr, mono.ptr, duration, 1, self._ir.ptr, self._ir_len, 2, 2, self._hist.ptr,
                             self._workspace.ptr), "pgx_convolve")
        self._last_render_end = start + duration
        return out

    def __repr__(self) -> str:
        return f"SpatialHRTF(azimuth={self.azimuth:.1f}, elevation={self.elevation:.1f})"


class SpatialPE(ProcessingElement):
    def __init__(self, source: ProcessingElement, *, method: SpatialMethod):
        if method is None:
            raise ValueError("SpatialPE: method is required")
        self._source = source
        self._method = method

    source = property(lambda self: self._source)
    method = property(lambda self: self._method)

    def inputs(self) -> list[ProcessingElement]:
        return [self._source, *self._method.inputs()]

    def is_pure(self) -> bool:
        return True          # declared pure by the reference (spatial_pe.py:629-632), HRTF tail notwithstanding

    def channel_count(self) -> int | None:
        return self._method.output_channels

    def _compute_extent(self) -> Extent:
        return self._source.extent()

    def _render(self, start: int, duration: int) -> Snippet:
        src = self._source.render(start, duration)
        return Snippet(start, self._method.render(src, start, duration, self._source.sample_rate or 0))

    def __repr__(self) -> str:
        return f"SpatialPE(source={type(self._source).__name__}, method={self._method})"
