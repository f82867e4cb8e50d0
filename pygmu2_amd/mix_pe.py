"""
MixPE: sum of N inputs in float32, in input order (mix_pe.py:47-153).

Inputs whose extent misses the requested window are skipped, as in the reference.
When every input is a structurally identical stateful voice graph the inputs are
rendered as ONE batched launch per PE level (pygmu2_amd.voice_bank) and summed by a
single kernel in the same float32 order -- same samples, far fewer launches.
Multi-GPU: pygmu2_amd.sharding.ShardedMixPE splits the inputs over ranks and reduces
the partial mixes with RCCL.
"""

from __future__ import annotations

import ctypes as C

from ._kernels import check, lib, new_output
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


class MixPE(ProcessingElement):
    _PASSES_BLOCKS = True              # look_ahead.py: inputs are pulled with the caller's (duration)
    _READ_AHEAD_SAFE = True

    def __init__(self, *inputs: ProcessingElement):
        if len(inputs) == 1 and isinstance(inputs[0], (list, tuple)):
            inputs = tuple(inputs[0])
        if len(inputs) < 2:
            raise ValueError("MixPE requires at least 2 inputs")
        self._inputs = list(inputs)
        self._bank = None            # lazily built voice bank (or False when not batchable)

    def inputs(self) -> list[ProcessingElement]:
        return self._inputs

    def is_pure(self) -> bool:
        return True

    def channel_count(self) -> int | None:
        return self._inputs[0].channel_count() if self._inputs else None

    def required_input_channels(self) -> int | None:
        return None

    def resolve_channel_count(self, input_channel_counts: list[int]) -> int:
        if not input_channel_counts:
            raise ValueError("MixPE has no inputs")
        first = input_channel_counts[0]
        for i, count in enumerate(input_channel_counts[1:], start=2):
            if count != first:
                raise ValueError(f"MixPE input channel mismatch: input 1 has {first} channels, "
                                 f"input {i} has {count} channels")
        return first

    def _compute_extent(self) -> Extent:
        ext = self._inputs[0].extent()
        for pe in self._inputs[1:]:
            ext = ext.union(pe.extent())
        return ext

    def _voice_bank(self):
        if self._bank is None:
            from .voice_bank import try_build_bank
            self._bank = try_build_bank(self._inputs) or False
            if self._bank and self.__dict__.get("_mix_windows"):
                self._bank.set_mix_windows()
        return self._bank

    def _read_ahead_condition(self) -> bool:
        # the skip rule below looks at the requested window; with bounded inputs a larger window changes it
        return all(pe.extent().start is None and pe.extent().end is None for pe in self._inputs)

    def _look_ahead_condition(self) -> bool:
        return not self._voice_bank()      # a bank keeps its voices' states in its own nodes: not snapshot

    def _render(self, start: int, duration: int) -> Snippet:
        bank = self._voice_bank()
        if bank:
            return bank.render_mix(start, duration)

        window = Extent(start, start + duration)
        snippets = [pe.render(start, duration) for pe in self._inputs
                    if pe.extent().intersects(window)]
        if not snippets:
            return Snippet.from_zeros(start, duration, self.channel_count() or 1)
        ch = snippets[0].channels
        for s in snippets[1:]:
            if s.channels != ch:
                raise ValueError(f"operands could not be broadcast together with shapes "
                                 f"({duration},{ch}) ({duration},{s.channels})")
        out = new_output(duration, ch)
        ptrs = (C.c_void_p * len(snippets))(*[s.dev.ptr for s in snippets])
        check(lib().pgx_mix_n(out.ptr, ptrs, len(snippets), duration * ch), "pgx_mix_n")
        return Snippet(start, out)

    def _on_start(self) -> None:
        if self._bank:
            self._bank.reset()

    _on_stop = _on_start

    def __repr__(self) -> str:
        return f"MixPE({', '.join(type(pe).__name__ for pe in self._inputs)})"
