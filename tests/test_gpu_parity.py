"""GPU: the HIP render path (pygmu2_amd, through the C ABI) against the golden vectors the
reference produced and against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star):
  * integer / index / copy PEs, float32-only arithmetic (GainPE, MixPE) on exact inputs,
    ADSR, gates, triggers, comb with exact control -> BIT-EXACT
  * float64 oscillators / filters / convolution       -> max|d| <= 1e-5 * max|ref| per block
    (peak-relative, SURVEY.md section 7 "parity metric"), with an absolute floor of 1e-7 for
    near-silent blocks.
"""

import json
import os

import numpy as np
import pytest

from oracle.graph_eval import run_case as oracle_run
from spec_build import run_case as hip_run

pytestmark = pytest.mark.gpu

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN_DIR, "cases.json")) as _f:
    CASES = json.load(_f)

REL_TOL = 1e-5      # of the block's peak |reference| (north_star: 1e-5 relative on float paths)
ABS_FLOOR = 1e-7

BIT_EXACT = {
    "constant_stereo", "identity_neg", "dirac_window", "array_zero", "array_hold_both",
    "crop_zero", "crop_hold", "crop_open_end", "gain_const_third",
    "periodic_gate", "periodic_gate_odd", "periodic_trigger",
    "periodic_gate_fm", "periodic_gate_pwm", "adsr_gate_fm",
    "adsr_full_cycle", "adsr_early_release_chunked", "adsr_bad_gate_values", "adsr_periodic_gate",
    "adsr_sustain_edges", "adsr_triggered", "adsr_triggered_retrigger",
    "comb_kat", "comb_high_freq", "comb_step",
    "transform_chain",
    "delay_int", "delay_int_negative", "delay_float_linear", "delay_float_cubic",
    "piecewise_step", "piecewise_linear", "piecewise_single_point", "piecewise_single_point_zero",
    "spatial_adapter_1_to_2", "spatial_adapter_2_to_1", "spatial_adapter_2_to_4", "spatial_adapter_4_to_2",
    "spatial_adapter_3_to_5", "spatial_adapter_5_to_2", "spatial_adapter_1_to_4", "spatial_adapter_2_to_2",
    "spatial_linear_scalar",
    "loop_count_crossfade", "loop_crossfade_clamped", "window_max", "window_min", "window_tiny",
}


def _compare(name, i, got, want, exact):
    assert got.dtype == np.float32 and got.shape == want.shape, (name, i, got.shape, want.shape)
    if exact:
        assert np.array_equal(got, want), (
            f"{name} block {i}: {int(np.sum(got != want))} of {want.size} samples differ, "
            f"max|d|={np.max(np.abs(got.astype(np.float64) - want))}")
        return
    assert np.all(np.isfinite(got)), f"{name} block {i}: non-finite output"
    peak = float(np.max(np.abs(want))) if want.size else 0.0
    err = float(np.max(np.abs(got.astype(np.float64) - want.astype(np.float64)))) if want.size else 0.0
    assert err <= REL_TOL * peak + ABS_FLOOR, (
        f"{name} block {i}: max|d|={err:.3e} > {REL_TOL:g}*peak({peak:.3e})")


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_hip_matches_reference_golden(case, golden_data):
    outs = hip_run(case)
    for i in case["keep"]:
        _compare(case["name"], i, outs[i], golden_data[f"{case['name']}/{i}"], case["name"] in BIT_EXACT)


@pytest.mark.parametrize("case", [c for c in CASES if len(c["keep"]) < len(c["blocks"])],
                         ids=lambda c: c["name"])
def test_hip_matches_oracle_on_unstored_blocks(case):
    """Blocks that were rendered but not stored in the fixture are checked against the oracle."""
    outs = hip_run(case)
    ref = oracle_run(case)
    for i in range(len(case["blocks"])):
        _compare(case["name"], i, outs[i], ref[i], case["name"] in BIT_EXACT)
