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
    name: str
    pe_class: str
    call_count: int = 0
    total_time_ns: int = 0
    samples: int = 0

    @property
    def total_time_ms(self) -> float:
        return self.total_time_ns / 1e6

    @property
    def avg_time_ms(self) -> float:
        return self.total_time_ms / self.call_count if self.call_count else 0.0


@dataclass
class ProfileReport:
    pe_profiles: dict = field(default_factory=dict)
    total_render_time_ns: int = 0
    total_samples: int = 0
    render_count: int = 0

    @property
    def total_render_time_ms(self) -> float:
        return self.total_render_time_ns / 1e6

    def realtime_ratio(self, sample_rate: int) -> float:
        if not self.total_render_time_ns:
            return 0.0
        return (self.total_samples / sample_rate) / (self.total_render_time_ns / 1e9)

    def summary(self, sample_rate: int = 44100) -> str:
        lines = [f"renders: {self.render_count}  samples: {self.total_samples}  "
                 f"time: {self.total_render_time_ms:.3f} ms  "
                 f"x{self.realtime_ratio(sample_rate):.1f} realtime"]
        for p in sorted(self.pe_profiles.values(), key=lambda q: -q.total_time_ns):
            lines.append(f"  {p.name:<40s} calls={p.call_count:<6d} total={p.total_time_ms:.3f} ms")
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
            t0 = time.perf_counter_ns()
            snippet = self._source.render(start, duration)
            self._output(snippet)
            dt = time.perf_counter_ns() - t0
            rep = self._profile_report
            rep.render_count += 1
            rep.total_render_time_ns += dt
            rep.total_samples += duration
            key = id(self._source)
            prof = rep.pe_profiles.get(key)
            if prof is None:
                cls = type(self._source).__name__
                prof = rep.pe_profiles[key] = PEProfile(name=f"{cls} (whole graph)", pe_class=cls)
            prof.call_count += 1
            prof.total_time_ns += dt
            prof.samples += duration
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
