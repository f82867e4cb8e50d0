"""
WavReaderPE: finite source backed by a WAV file, extent (0, frames), zeros outside
(wav_reader_pe.py:17-157).  The file is read once, uploaded as stored (int16 for PCM_16,
2 bytes per sample over PCIe) and converted on the device (pgx_pcm16_to_f32, libsndfile's
1/32768); every render is then a window copy out of HBM.
"""

from __future__ import annotations

import numpy as np

from ._kernels import DeviceBuffer, check, lib, new_output
from .extent import Extent
from .snippet import Snippet
from .source_pe import SourcePE
from .wav_io import WavInfo, read_frames, read_info


class WavReaderPE(SourcePE):
    _READ_AHEAD_SAFE = True

    def __init__(self, path: str):
        self._path = path
        self._info: WavInfo | None = None
        self._dev: DeviceBuffer | None = None

    path = property(lambda self: self._path)

    def _ensure_file_info(self) -> WavInfo:
        if self._info is None:
            self._info = read_info(self._path)
        return self._info

    @property
    def file_sample_rate(self) -> int | None:
        return self._ensure_file_info().sample_rate

    @property
    def sample_rate(self) -> int | None:
        if self._sample_rate is not None:
            return self._sample_rate
        return self.file_sample_rate

    def _on_start(self) -> None:
        self._ensure_file_info()

    def _compute_extent(self) -> Extent:
        return Extent(0, self._ensure_file_info().frames)

    def channel_count(self) -> int:
        return self._ensure_file_info().channels

    def _resident(self) -> DeviceBuffer:
        if self._dev is None:
            info = self._ensure_file_info()
            raw = read_frames(self._path, info, 0, info.frames)
            if info.format_tag == 3:
                self._dev = DeviceBuffer.from_host(np.ascontiguousarray(raw, dtype=np.float32))
            else:
                stored = DeviceBuffer.from_host(np.ascontiguousarray(raw, dtype=np.int16))
                self._dev = DeviceBuffer((info.frames, info.channels), np.float32)
                check(lib().pgx_pcm16_to_f32(self._dev.ptr, stored.ptr, info.frames * info.channels),
                      "pgx_pcm16_to_f32")
        return self._dev

    def _render(self, start: int, duration: int) -> Snippet:
        info = self._ensure_file_info()
        out = new_output(duration, info.channels, zero=info.frames == 0)
        if info.frames:
            check(lib().pgx_window_copy(out.ptr, start, duration, info.channels, self._resident().ptr, 0,
                                        info.frames, 0, 0), "pgx_window_copy")
        return Snippet(start, out)

    def __repr__(self) -> str:
        return f"WavReaderPE(path={self._path!r})"
