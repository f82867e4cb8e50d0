"""
WavWriterPE: a tap in the pull chain that records what flows through it.

Behaviour (reference wav_writer_pe.py:17-171): between Renderer.start() and stop() every block pulled
through the PE is appended to a WAV file, and the caller receives the source's Snippet itself; the file
takes the configured sample rate unless one is given, and the channel count of the source (or, for a
channel-agnostic source, of the first PE feeding it).

Device side: for the default subtype PCM_16 the block is quantised where it lives (pgx_f32_to_pcm16:
libsndfile's clipping conversion) so that 2 bytes per sample cross PCIe; FLOAT blocks are copied as they are.
"""

from __future__ import annotations

import numpy as np

from ._kernels import DeviceBuffer, check, lib
from .config import handle_error
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet
from .wav_io import WavFileWriter

_PCM16_TAG = 1          # WAVE_FORMAT_PCM in wav_io.WavFileWriter.tag


class WavWriterPE(ProcessingElement):
    def __init__(self, source: ProcessingElement, path: str, sample_rate: int | None = None,
                 subtype: str = "PCM_16"):
        self._source = source
        self._target = (path, sample_rate, subtype)
        self._sink: WavFileWriter | None = None
        self._count = 0

    # ------------------------------------------------------------------ description
    @property
    def path(self) -> str:
        return self._target[0]

    @property
    def frames_written(self) -> int:
        return self._count

    def inputs(self) -> list[ProcessingElement]:
        return [self._source]

    def is_pure(self) -> bool:
        return False                      # a side effect per pull: never rendered ahead, never shared

    def channel_count(self) -> int | None:
        return self._source.channel_count()

    def _compute_extent(self) -> Extent:
        return self._source.extent()

    def __repr__(self) -> str:
        path, _, subtype = self._target
        return f"WavWriterPE(source={type(self._source).__name__}, path={path!r}, subtype={subtype!r})"

    # ------------------------------------------------------------------ lifecycle
    def _file_channels(self) -> int | None:
        """Channels of the file: the source's, else those of the first PE behind a channel-agnostic source."""
        for pe in [self._source] + self._source.inputs()[:1]:
            count = pe.channel_count()
            if count is not None:
                return count
        return None

    def _on_start(self) -> None:
        path, rate, subtype = self._target
        channels = self._file_channels()
        if channels is None:
            handle_error(f"Cannot determine channel count for WavWriterPE. Source "
                         f"{type(self._source).__name__} returns None for channel_count().", fatal=True)
            return
        self._sink = WavFileWriter(path, rate or self.sample_rate, channels, subtype)
        self._count = 0

    def _on_stop(self) -> None:
        sink, self._sink = self._sink, None
        if sink is not None:
            sink.close()

    # ------------------------------------------------------------------ the tap
    def _append(self, block: Snippet) -> None:
        if self._sink.tag == _PCM16_TAG:
            pcm = DeviceBuffer((block.duration, block.channels), np.int16)
            check(lib().pgx_f32_to_pcm16(pcm.ptr, block.dev.ptr, block.duration * block.channels),
                  "pgx_f32_to_pcm16")
            self._sink.write(pcm.to_host())
        else:
            self._sink.write(block.data)
        self._count += block.duration

    def _render(self, start: int, duration: int) -> Snippet:
        block = self._source.render(start, duration)
        if self._sink is not None:
            self._append(block)
        return block
