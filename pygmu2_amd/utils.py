"""render_to_file (utils.py:34-62): render a finite PE into a WAV file through a NullRenderer
in one render call, like the reference."""

from __future__ import annotations

from .config import get_sample_rate
from .null_renderer import NullRenderer
from .processing_element import ProcessingElement
from .wav_writer_pe import WavWriterPE


def render_to_file(source: ProcessingElement, out_path: str, *, sample_rate: int | None = None, extent=None,
                   subtype: str = "PCM_16") -> None:
    sr = sample_rate if sample_rate is not None else get_sample_rate()
    if sr is None:
        raise RuntimeError("Sample rate not set. Call pg.set_sample_rate() or pass sample_rate.")
    if extent is None:
        extent = source.extent()
    if extent.start is None or extent.end is None:
        raise RuntimeError("Cannot render to file: source has infinite extent.")
    writer = WavWriterPE(source, out_path, sample_rate=int(sr), subtype=subtype)
    renderer = NullRenderer(sample_rate=int(sr))
    renderer.set_source(writer)
    with renderer:
        renderer.start()
        renderer.render(extent.start, extent.end - extent.start)
