"""
ReverbPE: convolution reverb with a wet/dry balance.

Behaviour (reference: reverb_pe.py:27-129): the source is pulled once per block (through a CachePE) and
feeds two branches that are summed -- the dry signal scaled by (1 - mix), and the source convolved with
the impulse response scaled by mix, optionally divided by the IR's energy norm.  `mix` is a number in
[0, 1] or a mono PE; for a PE the balance is computed per sample on the device.  Nothing is computed
here: the PE is a wiring of GainPE / ConvolvePE / MixPE, so every sample comes from their kernels.
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


def _validated_mix(mix):
    """A mono PE passes through; a number must lie in [0, 1] and comes back as float."""
    if isinstance(mix, ProcessingElement):
        channels = mix.channel_count()
        if channels is not None and int(channels) != 1:
            raise ValueError(f"mix PE must be mono, got {channels} channels")
        return mix
    value = float(mix)
    if value < 0.0 or value > 1.0:
        raise ValueError(f"mix must be in [0.0, 1.0], got {value}")
    return value


class ReverbPE(ProcessingElement):
    def __init__(self, source: ProcessingElement, ir: ProcessingElement, mix=0.5, *,
                 normalize_ir: bool = True, fft_size: int | None = None):
        balance = _validated_mix(mix)
        self._mix = mix
        self._ir = ir
        self._normalize_ir = bool(normalize_ir)
        self._fft_size = fft_size
        self._source = CachePE(source)                    # both branches read the same rendered block
        self._ir_energy = ConvolvePE.ir_energy_norm(ir) if self._normalize_ir else 1.0
        dry_level, wet_level = self._branch_levels(balance)
        self._wet_stream = ConvolvePE(self._source, ir, fft_size=fft_size)
        self._dry_gain = GainPE(self._source, gain=dry_level)
        self._wet_gain = GainPE(self._wet_stream, gain=wet_level)
        self._out = MixPE(self._dry_gain, self._wet_gain)

    def _branch_levels(self, balance):
        """(dry, wet) gains: numbers for a numeric balance, PEs for a PE balance."""
        if isinstance(balance, ProcessingElement):
            one_minus = MixPE(ConstantPE(1.0), GainPE(balance, gain=-1.0))
            wet = GainPE(balance, gain=1.0 / self._ir_energy) if self._normalize_ir else balance
            return one_minus, wet
        wet = balance / self._ir_energy if self._normalize_ir else balance
        return 1.0 - balance, wet

    @property
    def source(self):
        return self._source

    @property
    def ir(self):
        return self._ir

    @property
    def mix(self):
        return self._mix

    @property
    def ir_energy(self):
        return self._ir_energy

    # ------------------------------------------------------------------ PE contract: defer to the wiring
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
        def label(x):
            return type(x).__name__ if isinstance(x, ProcessingElement) else str(x)
        return (f"ReverbPE(source={label(self._source)}, ir={label(self._ir)}, mix={label(self._mix)}, "
                f"normalize_ir={self._normalize_ir}, fft_size={self._fft_size})")
