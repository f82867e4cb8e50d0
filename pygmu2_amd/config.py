"""
Global configuration: the process-wide sample rate and the error policy.

Mirrors the behaviour of the reference's config module (config.py:21-29, 32-109):
the sample rate must be set before any PE is constructed; errors are raised in STRICT
mode and logged in LENIENT mode unless marked fatal.
"""

from __future__ import annotations

import logging
from enum import Enum

_log = logging.getLogger("pygmu2_amd")

_sample_rate: int | None = None


def set_sample_rate(rate: int) -> None:
    global _sample_rate
    _sample_rate = int(rate)


def get_sample_rate() -> int | None:
    return _sample_rate


class ErrorMode(Enum):
    STRICT = "strict"
    LENIENT = "lenient"


_error_mode = ErrorMode.STRICT


def set_error_mode(mode: ErrorMode) -> None:
    global _error_mode
    _error_mode = mode


def get_error_mode() -> ErrorMode:
    return _error_mode


def handle_error(message: str, fatal: bool = False, error_mode: ErrorMode | None = None,
                 exception_class: type = RuntimeError) -> bool:
    """Raise `exception_class(message)` when fatal or STRICT; otherwise warn and return True."""
    mode = _error_mode if error_mode is None else error_mode
    if fatal or mode is ErrorMode.STRICT:
        raise exception_class(message)
    _log.warning(message)
    return True
