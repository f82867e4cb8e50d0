"""SourcePE: a PE without inputs (source_pe.py:16-52): pure by default, must state its
channel count."""

from __future__ import annotations

from abc import abstractmethod

from .processing_element import ProcessingElement


class SourcePE(ProcessingElement):
    def inputs(self) -> list[ProcessingElement]:
        return []

    def is_pure(self) -> bool:
        return True

    def required_input_channels(self) -> int | None:
        return None

    @abstractmethod
    def channel_count(self) -> int:
        ...
