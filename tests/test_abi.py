"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/pygmu_hip.h declares; without a GPU the compute entry points refuse to run
(no silent CPU fallback)."""

import ctypes
import os
import re

import pytest

from pygmu2_amd import device
from pygmu2_amd.build import LIB_PATH, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build()
    return device.load_library()


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "pygmu_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    declared = _header_symbols()
    assert len(declared) >= 40
    raw = ctypes.CDLL(LIB_PATH)
    missing = [s for s in declared if not hasattr(raw, s)]
    assert not missing, f"declared in pygmu_hip.h but not exported: {missing}"


def test_binding_covers_every_declared_symbol(lib):
    assert sorted(device.EXPORTED_SYMBOLS) == _header_symbols()
    assert lib.pgx_abi_version() == 1


def test_struct_mirrors_match_c_layout():
    assert device.SINE_PARAMS.itemsize == 24
    assert device.SINE_STATEFUL_PARAMS.itemsize == 32
    assert device.BIQUAD_VAR_PARAMS.itemsize == 32
    assert device.BLITSAW_PARAMS.itemsize == 32
    assert device.LADDER_PARAMS.itemsize == 40
    assert device.GATE_PARAMS.itemsize == 24
    assert device.ADSR_PARAMS.itemsize == 40


@pytest.mark.skipif(device.device_available(), reason="a GPU is present")
def test_no_cpu_fallback_without_gpu(lib):
    import pygmu2_amd as pg
    pg.set_sample_rate(44100)
    with pytest.raises(RuntimeError, match="no HIP device|GPU"):
        pg.SinePE(440.0).render(0, 16)
    # entry points refuse to run before pgx_init
    assert lib.pgx_fill(None, 16, 0.0) == -3
    assert b"pgx_init" in lib.pgx_last_error()
    # pure planning helpers work without a device
    assert lib.pgx_biquad_workspace_bytes(1, 1_000_000, 1, 0) > 0
    assert lib.pgx_biquad_workspace_bytes(512, 48_000, 1, 0) == 0
    assert lib.pgx_biquad_workspace_bytes(1, 1_000_000, 1, 512) == 0      # settled single launch
    assert lib.pgx_biquad_workspace_bytes(1, 1_000_000, 1, 1 << 20) > 0    # too slow a decay: exact pair
    assert lib.pgx_convolve_workspace_bytes(96_000, 65_536, 2) > 0
