"""
CompressorPE, LimiterPE, ExpanderPE: level detector + gain computer around one cached source.

Behaviour (reference: compressor_pe.py:21-325): each of the three is a fixed wiring -- the source is
rendered once per block (CachePE) and goes both to an EnvelopePE (side chain) and to a DynamicsPE that
applies the gain the envelope calls for.  What differs is the recipe: the compressor detects RMS by default
and compresses above the threshold with automatic make-up gain; the limiter is a compressor pinned to ratio
100, hard knee, peak detection, look-ahead and no make-up; the expander gates below the threshold with a
peak detector.  No kernels of their own: samples come from pgx_envelope and pgx_dynamics.
"""

from __future__ import annotations

from .cache_pe import CachePE
from .dynamics_pe import DynamicsMode, DynamicsPE
from .envelope_pe import DetectionMode, EnvelopePE
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


def _setting(name):
    """Read-only view of one entry of the processor's settings."""
    return property(lambda self: self._settings[name])


class _SideChainProcessor(ProcessingElement):
    """source -> CachePE -> { EnvelopePE , DynamicsPE(audio, envelope) }; the DynamicsPE is the output."""

    _PASSES_BLOCKS = True              # look_ahead.py: inputs are pulled with the caller's (duration)


    _LOOK_AHEAD_SAFE = True            # look_ahead.py: a wiring without state of its own

    def _wire(self, source, *, detector: dict, computer: dict, settings: dict) -> None:
        tap = CachePE(source)
        follower = EnvelopePE(tap, **detector)
        self._source = tap
        self._envelope_pe = follower
        self._dynamics_pe = DynamicsPE(tap, follower, **computer)
        self._settings = dict(settings)

    threshold = _setting("threshold")
    attack = _setting("attack")
    release = _setting("release")
    knee = _setting("knee")
    stereo_link = _setting("stereo_link")

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


class CompressorPE(_SideChainProcessor):
    AUTO = "auto"

    def __init__(self, source: ProcessingElement, threshold: float = -20.0, ratio: float = 4.0,
                 attack: float = 0.01, release: float = 0.1, knee: float = 6.0,
                 makeup_gain: float | str = "auto", lookahead: float = 0.0,
                 detection: DetectionMode = DetectionMode.RMS, stereo_link: bool = True):
        self._wire(source,
                   detector=dict(attack=attack, release=release, lookahead=lookahead, mode=detection),
                   computer=dict(threshold=threshold, ratio=ratio, knee=knee, makeup_gain=makeup_gain,
                                 mode=DynamicsMode.COMPRESS, stereo_link=stereo_link),
                   settings=dict(threshold=threshold, ratio=ratio, attack=attack, release=release, knee=knee,
                                 makeup_request=makeup_gain, lookahead=lookahead, detection=detection,
                                 stereo_link=stereo_link))

    ratio = _setting("ratio")
    lookahead = _setting("lookahead")
    detection = _setting("detection")

    @property
    def makeup_gain(self):
        return self._dynamics_pe.makeup_gain             # the resolved value when "auto" was asked for

    def __repr__(self) -> str:
        s = self._settings
        makeup = "auto" if s["makeup_request"] == self.AUTO else f"{self.makeup_gain:.1f}"
        return (f"CompressorPE(threshold={s['threshold']}, ratio={s['ratio']}, attack={s['attack']}, "
                f"release={s['release']}, knee={s['knee']}, makeup={makeup}, lookahead={s['lookahead']})")


class LimiterPE(CompressorPE):
    def __init__(self, source: ProcessingElement, ceiling: float = -1.0, attack: float = 0.0005,
                 release: float = 0.05, lookahead: float = 0.005, stereo_link: bool = True):
        CompressorPE.__init__(self, source, threshold=ceiling, ratio=100.0, attack=attack, release=release,
                              knee=0.0, makeup_gain=0.0, lookahead=lookahead, detection=DetectionMode.PEAK,
                              stereo_link=stereo_link)
        self._settings["ceiling"] = ceiling

    ceiling = _setting("ceiling")

    def __repr__(self) -> str:
        s = self._settings
        return f"LimiterPE(ceiling={s['ceiling']}, release={s['release']}, lookahead={s['lookahead']})"


class ExpanderPE(_SideChainProcessor):
    def __init__(self, source: ProcessingElement, threshold: float = -40.0, attack: float = 0.001,
                 release: float = 0.05, gate_range: float = -80.0, knee: float = 0.0, stereo_link: bool = True):
        self._wire(source,
                   detector=dict(attack=attack, release=release, mode=DetectionMode.PEAK),
                   computer=dict(threshold=threshold, ratio=1.0, knee=knee, makeup_gain=0.0,
                                 mode=DynamicsMode.GATE, stereo_link=stereo_link, gate_range=gate_range),
                   settings=dict(threshold=threshold, attack=attack, release=release, knee=knee,
                                 gate_range=gate_range, stereo_link=stereo_link))

    gate_range = _setting("gate_range")

    def __repr__(self) -> str:
        s = self._settings
        return (f"ExpanderPE(threshold={s['threshold']}, attack={s['attack']}, release={s['release']}, "
                f"range={s['gate_range']})")
