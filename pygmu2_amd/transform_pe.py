"""
TransformPE: apply an element-wise function to a source (transform_pe.py:19-152).

`func` is a descriptor from pygmu2_amd.transforms (or np.abs / np.tanh / np.sqrt /
np.square, which are recognised): the whole chain runs in one device kernel in float64
and rounds to float32 once, like the reference's  func(data.astype(float64)).astype(float32).

Any other callable is the user's own host code.  It is honoured the way the reference
does it, on a host copy of the block (device -> host -> func -> device), with the
reference's shape repair (transform_pe.py:136-147); that crossing is the callable's cost,
not a fallback of this library.
"""

from __future__ import annotations

from typing import Callable

import numpy as np

from . import device as _dev
from . import transforms as _tf
from ._kernels import DeviceBuffer, check, lib, new_output
from .extent import Extent
from .processing_element import ProcessingElement
from .snippet import Snippet


class TransformPE(ProcessingElement):
    _PASSES_BLOCKS = True              # look_ahead.py: inputs are pulled with the caller's (duration)
    _READ_AHEAD_SAFE = True            # named element-wise chains only (see the condition)

    def _read_ahead_condition(self) -> bool:
        return self._lowered is not None

    def __init__(self, source: ProcessingElement, func: Callable[[np.ndarray], np.ndarray],
                 name: str | None = None):
        self._source = source
        self._func = func
        self._name = name or getattr(func, "__name__", "transform")
        self._lowered = _tf.lower(func)
        self._ops_dev: DeviceBuffer | None = None
        self._nops = 0

    source = property(lambda self: self._source)
    func = property(lambda self: self._func)
    name = property(lambda self: self._name)

    def inputs(self) -> list[ProcessingElement]:
        return [self._source]

    def is_pure(self) -> bool:
        return True

    def channel_count(self) -> int | None:
        return self._source.channel_count()

    def _compute_extent(self) -> Extent:
        return self._source.extent()

    def _render(self, start: int, duration: int) -> Snippet:
        src = self._source.render(start, duration)
        if self._lowered is None:
            return self._render_host_callable(start, src)
        if self._ops_dev is None:
            ops = self._lowered.ops()
            table = np.zeros(max(len(ops), 1), dtype=_dev.TRANSFORM_OP)
            for i, (code, p0, p1) in enumerate(ops):
                table[i] = (code, 0, p0, p1)
            self._ops_dev = DeviceBuffer.from_host(table.view(np.uint8))
            self._nops = len(ops)
        out = new_output(duration, src.channels)
        check(lib().pgx_transform(out.ptr, src.dev.ptr, duration * src.channels, self._ops_dev.ptr,
                                  self._nops), "pgx_transform")
        return Snippet(start, out)

    def _render_host_callable(self, start: int, src: Snippet) -> Snippet:
        data = src.data.astype(np.float64)
        res = np.asarray(self._func(data))
        if data.ndim == 2 and res.ndim == 1:
            res = res.reshape(-1, data.shape[1])
        elif data.ndim == 2 and res.ndim == 2 and res.shape[1] != data.shape[1]:
            res = np.broadcast_to(res, data.shape) if res.shape[1] == 1 else res[:, :data.shape[1]]
        return Snippet(start, np.ascontiguousarray(res.astype(np.float32)))

    def __repr__(self) -> str:
        return f"TransformPE(source={type(self._source).__name__}, func={self._name})"
