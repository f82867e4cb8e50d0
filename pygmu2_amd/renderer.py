"""
Renderer: validates a PE graph, drives its lifecycle and pulls Snippets from the root.

Behaviour follows the reference's Renderer (renderer.py:128-513):
  set_source -> graph validation (a non-pure PE may feed only one sink; channel counts
                must resolve), remembers the output channel count
  start      -> on_start() once per node, inputs before consumers
  render     -> root.render(start, duration >= 1) then _output(snippet)
  stop       -> on_stop() once per node, consumers before inputs; idempotent
Profiling attributes whole-graph wall time to the root PE, like the reference's
approximation (renderer.py:515-562); per-kernel evidence comes from rocprofv3.
"""

from __future__ import annotations

import logging
import time
from abc import ABC, abstractmethod
from dataclasses import dataclass, field

from .config import handle_error
from .processing_element import ProcessingElement
from .snippet import Snippet

_log = logging.getLogger("pygmu2_amd.renderer")


@dataclass
class PEProfile:
    """Timing of one ProcessingElement (renderer.py:25-62: same fields and derived figures)."""
    pe_class: str
    pe_id: int
    render_count: int = 0
    total_time_ns: int = 0
    total_samples: int = 0
    min_time_ns: int = 0
    max_time_ns: int = 0

    @property
    def total_time_ms(self) -> float:
        return self.total_time_ns / 1_000_000

    @property
    def avg_time_ms(self) -> float:
        return self.total_time_ms / self.render_count if self.render_count else 0.0

    @property
    def samples_per_second(self) -> float:
        return self.total_samples / (self.total_time_ns / 1_000_000_000) if self.total_time_ns else 0.0

    def realtime_ratio(self, sample_rate: int = 44100) -> float:
        if self.total_time_ns == 0:
            return 0.0
        return (self.total_samples / sample_rate) * 1_000_000_000 / self.total_time_ns


@dataclass
class ProfileReport:
    """Profile of a render session (renderer.py:65-127).  On the device a block is a pipeline of
    asynchronous launches, so the unit that can be timed honestly is the whole graph per render call (the
    stream is synchronised around it while profiling); per-kernel times are rocprofv3's job."""
    pe_profiles: dict = field(default_factory=dict)
    total_render_time_ns: int = 0
    total_output_time_ns: int = 0
    total_samples: int = 0
    render_calls: int = 0

    def add_pe_timing(self, pe: ProcessingElement, time_ns: int, samples: int) -> None:
        key = id(pe)
        prof = self.pe_profiles.get(key)
        if prof is None:
            prof = self.pe_profiles[key] = PEProfile(pe_class=type(pe).__name__, pe_id=key, min_time_ns=time_ns,
                                                     max_time_ns=time_ns)
        prof.render_count += 1
        prof.total_time_ns += time_ns
        prof.total_samples += samples
        prof.min_time_ns = min(prof.min_time_ns, time_ns)
        prof.max_time_ns = max(prof.max_time_ns, time_ns)

    def summary(self, sample_rate: int = 44100) -> str:
        lines = ["=" * 70, "RENDER PROFILE REPORT", "=" * 70,
                 f"Total render calls: {self.render_calls}", f"Total samples: {self.total_samples:,}",
                 f"Total render time: {self.total_render_time_ns / 1_000_000:.2f} ms",
                 f"Total output time: {self.total_output_time_ns / 1_000_000:.2f} ms"]
        if self.total_render_time_ns > 0:
            ratio = (self.total_samples / sample_rate) * 1_000_000_000 / self.total_render_time_ns
            lines.append(f"Realtime ratio: {ratio:.1f}x (>1.0x is faster than realtime)")
        lines += ["", "PER-PE BREAKDOWN (sorted by total time):", "-" * 70,
                  f"{'PE Class':<20} {'Calls':>8} {'Total ms':>10} {'Avg ms':>10} {'Samples/s':>12}", "-" * 70]
        for p in sorted(self.pe_profiles.values(), key=lambda q: q.total_time_ns, reverse=True):
            lines.append(f"{p.pe_class:<20} {p.render_count:>8} {p.total_time_ms:>10.2f} {p.avg_time_ms:>10.4f} "
                         f"{p.samples_per_second:>12,.0f}")
        lines.append("=" * 70)
        return "\n".join(lines)


class Renderer(ABC):
    def __init__(self, sample_rate: int = 44100):
        self._sample_rate = sample_rate
        self._source: ProcessingElement | None = None
        self._channel_count: int | None = None
        self._started = False
        self._profiling = False
        self._profile_report: ProfileReport | None = None
        self._pe_list: list[ProcessingElement] = []

    sample_rate = property(lambda self: self._sample_rate)
    source = property(lambda self: self._source)
    channel_count = property(lambda self: self._channel_count)
    started = property(lambda self: self._started)
    profiling = property(lambda self: self._profiling)

    # ------------------------------------------------------------------ profiling
    def enable_profiling(self) -> None:
        self._profiling = True
        self._profile_report = ProfileReport()

    def disable_profiling(self) -> None:
        self._profiling = False

    def get_profile_report(self) -> ProfileReport | None:
        return self._profile_report

    def print_profile_report(self) -> None:
        if self._profile_report is None:
            print("No profile data available. Call enable_profiling() first.")
        else:
            print(self._profile_report.summary(self._sample_rate))

    # ------------------------------------------------------------------ lifecycle
    def set_source(self, source: ProcessingElement) -> None:
        if self._started and handle_error("Cannot set source while started. Call stop() first."):
            return
        self._channel_count = self._validate_graph(source, {})
        self._source = source
        self._pe_list = []
        self._walk(source, set(), self._pe_list.append, post_order=True)

    def start(self) -> None:
        if self._source is None:
            handle_error("No source set. Call set_source() first.", fatal=True)
            return
        if self._started and handle_error("Already started. Call stop() first."):
            return
        self._walk(self._source, set(), lambda pe: pe.on_start(), post_order=True)
        self._started = True

    def stop(self) -> None:
        if not self._started:
            return
        if self._source is not None:
            self._walk(self._source, set(), lambda pe: pe.on_stop(), post_order=False)
        self._started = False

    def render(self, start: int, duration: int) -> None:
        if self._source is None:
            handle_error("No source set. Call set_source() first.", fatal=True)
            return
        if not self._started:
            handle_error("Not started. Call start() first.", fatal=True)
            return
        if duration < 1:
            handle_error("Renderer.render() requires duration >= 1 to prevent infinite loops.",
                         fatal=True, exception_class=ValueError)
            return
        if self._profiling and self._profile_report is not None:
            from . import device as _dev

            def settle():                    # device work is asynchronous: drain the stream around the render
                if _dev._initialised:
                    _dev.synchronize()

            rep = self._profile_report
            settle()
            t0 = time.perf_counter_ns()
            snippet = self._source.render(start, duration)
            settle()
            t1 = time.perf_counter_ns()
            self._output(snippet)
            t2 = time.perf_counter_ns()
            rep.render_calls += 1
            rep.total_samples += duration
            rep.total_render_time_ns += t1 - t0
            rep.total_output_time_ns += t2 - t1
            rep.add_pe_timing(self._source, t1 - t0, duration)      # the root stands for its whole graph
        else:
            self._output(self._source.render(start, duration))

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        self.stop()
        return False

    @abstractmethod
    def _output(self, snippet: Snippet) -> None:
        ...

    # ------------------------------------------------------------------ graph walks
    def _validate_graph(self, pe: ProcessingElement, seen: dict) -> int:
        key = id(pe)
        if key in seen:
            if not pe.is_pure():
                raise ValueError(f"{type(pe).__name__} is not pure but has multiple sinks. "
                                 f"Stateful PEs can only connect to one downstream PE.")
            return seen[key]
        ins = pe.inputs()
        in_channels = [self._validate_graph(child, seen) for child in ins]
        need = pe.required_input_channels()
        if need is not None:
            for child, got in zip(ins, in_channels):
                if got != need:
                    raise ValueError(f"{type(pe).__name__} requires {need} channel(s), "
                                     f"but {type(child).__name__} outputs {got}")
        out = pe.channel_count()
        if out is None:
            if not in_channels:
                raise ValueError(f"{type(pe).__name__} has no inputs but channel_count() is None")
            out = pe.resolve_channel_count(in_channels)
        seen[key] = out
        return out

    def _walk(self, pe: ProcessingElement, visited: set, visit, post_order: bool) -> None:
        key = id(pe)
        if key in visited:
            return
        visited.add(key)
        if not post_order:
            visit(pe)
        for child in pe.inputs():
            self._walk(child, visited, visit, post_order)
        if post_order:
            visit(pe)
