"""
WavWriterPE: pass-through PE that appends every rendered block to a WAV file
(wav_writer_pe.py:17-171).  PCM_16 blocks are quantised on the device (pgx_f32_to_pcm16,
libsndfile's rule) so the device->host copy carries 2 bytes per sample; the Snippet handed
on is the source's, untouched.
"""

from __future__ import annotations

import numpy as np

from ._kernels import DeviceBuffer, check, lib
from .config import handle_error
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet
from .wav_io import WavFileWriter


class WavWriterPE(ProcessingElement):
    def __init__(self, source: ProcessingElement, path: str, sample_rate: int | None = None,
                 subtype: str = "PCM_16"):
        self._source = source
        self._path = path
        self._output_sample_rate = sample_rate
        self._subtype = subtype
        self._file: WavFileWriter | None = None
        self._frames_written = 0

    path = property(lambda self: self._path)
    frames_written = property(lambda self: self._frames_written)

    def inputs(self) -> list[ProcessingElement]:
        return [self._source]

    def is_pure(self) -> bool:
        return False

    def channel_count(self) -> int | None:
        return self._source.channel_count()

    def _on_start(self) -> None:
        rate = self._output_sample_rate or self.sample_rate
        channels = self._source.channel_count()
        if channels is None:
            ins = self._source.inputs()
            if ins:
                channels = ins[0].channel_count()
        if channels is None:
            handle_error(f"Cannot determine channel count for WavWriterPE. Source "
                         f"{type(self._source).__name__} returns None for channel_count().", fatal=True)
            return
        self._file = WavFileWriter(self._path, rate, channels, self._subtype)
        self._frames_written = 0

    def _on_stop(self) -> None:
        if self._file is not None:
            self._file.close()
            self._file = None

    def _render(self, start: int, duration: int) -> Snippet:
        snippet = self._source.render(start, duration)
        if self._file is not None:
            if self._file.tag == 1:
                pcm = DeviceBuffer((snippet.duration, snippet.channels), np.int16)
                check(lib().pgx_f32_to_pcm16(pcm.ptr, snippet.dev.ptr, snippet.duration * snippet.channels),
                      "pgx_f32_to_pcm16")
                self._file.write(pcm.to_host())
            else:
                self._file.write(snippet.data)
            self._frames_written += snippet.duration
        return snippet

    def _compute_extent(self) -> Extent:
        return self._source.extent()

    def __repr__(self) -> str:
        return f"WavWriterPE(source={type(self._source).__name__}, path={self._path!r}, subtype={self._subtype!r})"
