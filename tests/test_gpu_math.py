"""GPU: accuracy of the library's float64 sin/cos (Cody-Waite + Taylor, pgx_common.h) against
numpy's -- the routine behind every oscillator phase and every RBJ coefficient."""

import numpy as np
import pytest

from pygmu2_amd import device

pytestmark = pytest.mark.gpu


def _eval(x):
    lib = device.ensure_init()
    xd = device.DeviceBuffer.from_host(x)
    s = device.DeviceBuffer(x.shape, np.float64)
    c = device.DeviceBuffer(x.shape, np.float64)
    device.check(lib.pgx_selftest_sincos(s.ptr, c.ptr, xd.ptr, x.size))
    return s.to_host(), c.to_host()


def test_sincos_accuracy_over_oscillator_ranges():
    rng = np.random.default_rng(7)
    x = np.concatenate([
        rng.uniform(-np.pi, np.pi, 200_000),                 # theta = pi*phase
        rng.uniform(0, 2500.0, 200_000),                     # M*theta, M up to ~800
        rng.uniform(-3.0e6, 3.0e6, 200_000),                 # long-running sine phase
        np.pi * rng.integers(-1000, 1000, 50_000) + rng.uniform(-1e-6, 1e-6, 50_000),   # near zeros of sin
        np.pi * (rng.integers(-1000, 1000, 50_000) + 0.5) + rng.uniform(-1e-6, 1e-6, 50_000),
        rng.uniform(3.0e6, 1e9, 10_000),                     # beyond the fast range: ocml fallback
        np.array([0.0, -0.0, 1e-300, np.pi, -np.pi, 0.5 * np.pi]),
    ])
    s, c = _eval(x)
    assert not np.any(np.isnan(c)), "pgx_sin and pgx_sincos disagree"
    ws, wc = np.sin(x), np.cos(x)
    # sine: relative accuracy (in ulps of the result) except where |sin| is below 1e-9 of the argument scale
    ulp = np.spacing(np.abs(ws))
    big = np.abs(ws) > 1e-12
    assert np.max(np.abs(s - ws)[big] / ulp[big]) <= 4.0
    assert np.max(np.abs(s - ws)) <= 4e-16
    assert np.max(np.abs(c - wc)) <= 4e-16


def test_tanh_accuracy_over_ladder_feedback_range():
    """pgx_tanh (the ladder's feedback nonlinearity): <= 4 ulp for |x| >= 0.25, absolute <= 2.5e-16 below."""
    lib = device.ensure_init()
    rng = np.random.default_rng(11)
    x = np.concatenate([
        rng.uniform(-0.25, 0.25, 200_000), rng.uniform(-4.0, 4.0, 200_000), rng.uniform(-45.0, 45.0, 100_000),
        10.0 ** rng.uniform(-12, -1, 50_000) * rng.choice([-1.0, 1.0], 50_000),
        np.array([0.0, -0.0, 1e-300, -1e-300, 0.25, -0.25, 19.0, 40.0, 1e300, -1e300, np.inf, -np.inf]),
    ])
    xd = device.DeviceBuffer.from_host(x)
    out = device.DeviceBuffer(x.shape, np.float64)
    device.check(lib.pgx_selftest_tanh(out.ptr, xd.ptr, x.size))
    got, want = out.to_host(), np.tanh(x)
    assert np.all(np.isfinite(got))
    assert np.array_equal(np.signbit(got), np.signbit(want))
    err = np.abs(got - want)
    big = np.abs(x) >= 0.25
    assert np.max(err[big] / np.spacing(np.abs(want[big]))) <= 4.0
    assert np.max(err[~big]) <= 2.5e-16
    assert np.all(np.abs(got) <= 1.0)
    nan_in = device.DeviceBuffer.from_host(np.array([np.nan]))
    nan_out = device.DeviceBuffer((1,), np.float64)
    device.check(lib.pgx_selftest_tanh(nan_out.ptr, nan_in.ptr, 1))
    assert np.isnan(nan_out.to_host()[0])
