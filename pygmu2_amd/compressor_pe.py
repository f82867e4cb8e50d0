"""
CompressorPE, LimiterPE, ExpanderPE: CachePE(source) feeding an EnvelopePE and a DynamicsPE
(compressor_pe.py:21-325).  No kernels of their own: the envelope follower and the gain computer are
pgx_envelope and pgx_dynamics.
"""

from __future__ import annotations

from .cache_pe import CachePE
from .dynamics_pe import DynamicsMode, DynamicsPE
from .envelope_pe import DetectionMode, EnvelopePE
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


class _DynamicsProcessorPE(ProcessingElement):
    def __init__(self, cached_source, envelope_pe, dynamics_pe, *, threshold, attack, release, knee, stereo_link):
        self._source = cached_source
        self._envelope_pe = envelope_pe
        self._dynamics_pe = dynamics_pe
        self._threshold = threshold
        self._attack = attack
        self._release = release
        self._knee = knee
        self._stereo_link = stereo_link

    threshold = property(lambda self: self._threshold)
    attack = property(lambda self: self._attack)
    release = property(lambda self: self._release)
    knee = property(lambda self: self._knee)
    stereo_link = property(lambda self: self._stereo_link)

    def inputs(self) -> list[ProcessingElement]:
        return [self._dynamics_pe]

    def is_pure(self) -> bool:
        return False

    def channel_count(self) -> int | None:
        return self._dynamics_pe.channel_count()

    def _compute_extent(self) -> Extent:
        return self._dynamics_pe.extent()

    def _render(self, start: int, duration: int) -> Snippet:
        return self._dynamics_pe.render(start, duration)


class CompressorPE(_DynamicsProcessorPE):
    AUTO = "auto"

    def __init__(self, source: ProcessingElement, threshold: float = -20.0, ratio: float = 4.0,
                 attack: float = 0.01, release: float = 0.1, knee: float = 6.0,
                 makeup_gain: float | str = "auto", lookahead: float = 0.0,
                 detection: DetectionMode = DetectionMode.RMS, stereo_link: bool = True):
        cached = CachePE(source)
        envelope_pe = EnvelopePE(cached, attack=attack, release=release, lookahead=lookahead, mode=detection)
        dynamics_pe = DynamicsPE(cached, envelope_pe, threshold=threshold, ratio=ratio, knee=knee,
                                 makeup_gain=makeup_gain, mode=DynamicsMode.COMPRESS, stereo_link=stereo_link)
        super().__init__(cached, envelope_pe, dynamics_pe, threshold=threshold, attack=attack, release=release,
                         knee=knee, stereo_link=stereo_link)
        self._ratio = ratio
        self._makeup_gain = makeup_gain
        self._lookahead = lookahead
        self._detection = detection

    ratio = property(lambda self: self._ratio)
    makeup_gain = property(lambda self: self._dynamics_pe.makeup_gain)
    lookahead = property(lambda self: self._lookahead)
    detection = property(lambda self: self._detection)

    def __repr__(self) -> str:
        makeup_str = "auto" if self._makeup_gain == self.AUTO else f"{self.makeup_gain:.1f}"
        return (f"CompressorPE(threshold={self._threshold}, ratio={self._ratio}, attack={self._attack}, "
                f"release={self._release}, knee={self._knee}, makeup={makeup_str}, lookahead={self._lookahead})")


class LimiterPE(CompressorPE):
    def __init__(self, source: ProcessingElement, ceiling: float = -1.0, attack: float = 0.0005,
                 release: float = 0.05, lookahead: float = 0.005, stereo_link: bool = True):
        super().__init__(source, threshold=ceiling, ratio=100.0, attack=attack, release=release, knee=0.0,
                         makeup_gain=0.0, lookahead=lookahead, detection=DetectionMode.PEAK,
                         stereo_link=stereo_link)
        self._ceiling = ceiling

    ceiling = property(lambda self: self._ceiling)

    def __repr__(self) -> str:
        return f"LimiterPE(ceiling={self._ceiling}, release={self._release}, lookahead={self._lookahead})"


class ExpanderPE(_DynamicsProcessorPE):
    def __init__(self, source: ProcessingElement, threshold: float = -40.0, attack: float = 0.001,
                 release: float = 0.05, gate_range: float = -80.0, knee: float = 0.0, stereo_link: bool = True):
        cached = CachePE(source)
        envelope_pe = EnvelopePE(cached, attack=attack, release=release, mode=DetectionMode.PEAK)
        dynamics_pe = DynamicsPE(cached, envelope_pe, threshold=threshold, ratio=1.0, knee=knee, makeup_gain=0.0,
                                 mode=DynamicsMode.GATE, stereo_link=stereo_link, gate_range=gate_range)
        super().__init__(cached, envelope_pe, dynamics_pe, threshold=threshold, attack=attack, release=release,
                         knee=knee, stereo_link=stereo_link)
        self._range = gate_range

    gate_range = property(lambda self: self._range)

    def __repr__(self) -> str:
        return (f"ExpanderPE(threshold={self._threshold}, attack={self._attack}, release={self._release}, "
                f"range={self._range})")
