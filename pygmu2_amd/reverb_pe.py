"""
ReverbPE: convolution reverb with a wet/dry mix (reverb_pe.py:27-129) -- a composition of
the hot-path PEs:  MixPE(GainPE(src, 1 - mix), GainPE(ConvolvePE(src, ir), mix / ir_energy)),
the source pulled once through a CachePE.
"""

from __future__ import annotations

from .cache_pe import CachePE
from .constant_pe import ConstantPE
from .convolve_pe import ConvolvePE
from .extent import Extent
from .gain_pe import GainPE
from .mix_pe import MixPE
from .processing_element import ProcessingElement
from .snippet import Snippet


class ReverbPE(ProcessingElement):
    def __init__(self, source: ProcessingElement, ir: ProcessingElement, mix=0.5, *,
                 normalize_ir: bool = True, fft_size: int | None = None):
        self._source = CachePE(source)
        self._ir = ir
        self._mix = mix
        self._normalize_ir = bool(normalize_ir)
        self._fft_size = fft_size
        if isinstance(mix, ProcessingElement):
            mix_ch = mix.channel_count()
            if mix_ch is not None and int(mix_ch) != 1:
                raise ValueError(f"mix PE must be mono, got {mix_ch} channels")
        else:
            mix = float(mix)
            if not (0.0 <= mix <= 1.0):
                raise ValueError(f"mix must be in [0.0, 1.0], got {mix}")
        self._ir_energy = ConvolvePE.ir_energy_norm(self._ir) if self._normalize_ir else 1.0
        self._wet_stream = ConvolvePE(self._source, self._ir, fft_size=self._fft_size)
        if isinstance(self._mix, ProcessingElement):
            dry_gain = MixPE(ConstantPE(1.0), GainPE(self._mix, gain=-1.0))
            wet_gain = self._mix
            if self._normalize_ir:
                wet_gain = GainPE(wet_gain, gain=(1.0 / self._ir_energy))
        else:
            dry_gain = 1.0 - float(self._mix)
            wet_gain = float(self._mix)
            if self._normalize_ir:
                wet_gain = wet_gain / self._ir_energy
        self._dry_gain = GainPE(self._source, gain=dry_gain)
        self._wet_gain = GainPE(self._wet_stream, gain=wet_gain)
        self._out = MixPE(self._dry_gain, self._wet_gain)

    source = property(lambda self: self._source)
    ir = property(lambda self: self._ir)
    mix = property(lambda self: self._mix)
    ir_energy = property(lambda self: self._ir_energy)

    def inputs(self) -> list[ProcessingElement]:
        return [self._out]

    def is_pure(self) -> bool:
        return False

    def channel_count(self) -> int | None:
        return self._out.channel_count()

    def _compute_extent(self) -> Extent:
        return self._out.extent()

    def _render(self, start: int, duration: int) -> Snippet:
        return self._out.render(start, duration)

    def __repr__(self) -> str:
        mix = type(self._mix).__name__ if isinstance(self._mix, ProcessingElement) else str(self._mix)
        return (f"ReverbPE(source={type(self._source).__name__}, ir={type(self._ir).__name__}, mix={mix}, "
                f"normalize_ir={self._normalize_ir}, fft_size={self._fft_size})")
