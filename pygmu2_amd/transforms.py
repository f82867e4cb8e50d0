"""
Named element-wise transforms for TransformPE.

The reference's TransformPE (transform_pe.py:19-66) takes an arbitrary Python callable
over numpy arrays.  A callable cannot cross the C ABI, so the common shapes are offered as
small descriptor objects that TransformPE lowers to one device kernel (pgx_transform).
Every descriptor is also a plain numpy callable with the same float64 arithmetic, so the
same object can be handed to the reference's TransformPE.

    Affine(scale, offset)   offset + scale * x          Abs()        |x|
    Clip(lo, hi)            np.clip(x, lo, hi)          Tanh()       np.tanh(x)
    Sqrt()                  x ** 0.5                    OneMinus()   1.0 - x
    Square()                x ** 2                      Chain(a, b, ...)  b(a(x)) ...
"""

from __future__ import annotations

import numpy as np

AFFINE, CLIP, SQRT, SQUARE, ABS, TANH, ONE_MINUS = range(7)


class DeviceTransform:
    """Base class: `ops()` lists (code, p0, p1) triples in application order."""

    def ops(self) -> list[tuple[int, float, float]]:
        raise NotImplementedError

    def __call__(self, v):
        raise NotImplementedError

    @property
    def __name__(self) -> str:        # TransformPE's default display name
        return type(self).__name__.lower()


class Affine(DeviceTransform):
    def __init__(self, scale: float = 1.0, offset: float = 0.0):
        self.scale, self.offset = float(scale), float(offset)

    def ops(self):
        return [(AFFINE, self.scale, self.offset)]

    def __call__(self, v):
        return self.offset + self.scale * v


class Clip(DeviceTransform):
    def __init__(self, lo: float, hi: float):
        if lo > hi:
            raise ValueError(f"Clip: lo ({lo}) > hi ({hi})")
        self.lo, self.hi = float(lo), float(hi)

    def ops(self):
        return [(CLIP, self.lo, self.hi)]

    def __call__(self, v):
        return np.clip(v, self.lo, self.hi)


class _Unary(DeviceTransform):
    code = -1

    def ops(self):
        return [(self.code, 0.0, 0.0)]


class Sqrt(_Unary):
    code = SQRT

    def __call__(self, v):
        return v ** 0.5


class Square(_Unary):
    code = SQUARE

    def __call__(self, v):
        return v ** 2


class Abs(_Unary):
    code = ABS

    def __call__(self, v):
        return np.abs(v)


class Tanh(_Unary):
    code = TANH

    def __call__(self, v):
        return np.tanh(v)


class OneMinus(_Unary):
    code = ONE_MINUS

    def __call__(self, v):
        return 1.0 - v


class Chain(DeviceTransform):
    def __init__(self, *steps: DeviceTransform):
        for s in steps:
            if not isinstance(s, DeviceTransform):
                raise TypeError(f"Chain takes DeviceTransform steps, got {type(s).__name__}")
        self.steps = steps

    def ops(self):
        return [op for s in self.steps for op in s.ops()]

    def __call__(self, v):
        for s in self.steps:
            v = s(v)
        return v


# numpy callables that mean the same thing as a descriptor
_NUMPY_EQUIVALENTS = {np.abs: Abs, np.absolute: Abs, np.fabs: Abs, np.tanh: Tanh, np.sqrt: Sqrt,
                      np.square: Square}


def lower(func):
    """DeviceTransform for `func`, or None when it is an opaque Python callable."""
    if isinstance(func, DeviceTransform):
        return func
    try:
        cls = _NUMPY_EQUIVALENTS.get(func)
    except TypeError:                 # unhashable callable
        cls = None
    return cls() if cls is not None else None


def from_spec(ops) -> Chain:
    """[(name, *params), ...] (the golden-case notation of oracle/golden_cases.py) -> Chain."""
    table = {"affine": Affine, "clip": Clip, "sqrt": Sqrt, "square": Square, "abs": Abs, "tanh": Tanh,
             "one_minus": OneMinus}
    return Chain(*(table[op[0]](*op[1:]) for op in ops))
