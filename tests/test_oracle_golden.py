"""CPU: the oracle (oracle/pe_oracle.py + oracle/seq_kernels.c) against the golden vectors
that oracle/gen_golden.py produced with the reference's own classes.

The oracle calls the reference's primitives in the reference's order, so the bar is
bit-exact equality on every stored block of every case."""

import json
import os

import numpy as np
import pytest

from oracle import pe_oracle as O
from oracle.graph_eval import run_case

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN_DIR, "cases.json")) as _f:
    _CASES = json.load(_f)


@pytest.mark.parametrize("case", _CASES, ids=[c["name"] for c in _CASES])
def test_oracle_matches_reference_golden(case, golden_data):
    outs = run_case(case)
    for i in case["keep"]:
        want = golden_data[f"{case['name']}/{i}"]
        got = outs[i]
        assert got.dtype == np.float32 and got.shape == want.shape
        assert np.array_equal(got, want), (
            f"{case['name']} block {i}: max|d|={np.max(np.abs(got.astype(np.float64) - want))}")


def test_known_answers_from_survey():
    """KATs quoted in SURVEY.md section 8(a) (values printed by the reference)."""
    y = O.sine_pure(0, 5, 440.0, sr=44100)[:, 0]
    assert np.allclose(y, [0, 0.0626483262, 0.12505053, 0.186961442, 0.248137847], atol=1e-9)
    b0, b1, b2, a1, a2 = (float(v[0]) for v in O.biquad_coeffs(np.array([1000.0]), np.array([0.707]),
                                                               "lowpass", 0.0, 44100))
    assert (b0, b1, a1, a2) == (0.004603935028493071, 0.009207870056986141,
                                -1.799071616595651, 0.8174873567096231)
    st = O.biquad_state(1)
    y = O.biquad_const(st, O.sine_pure(0, 5, 440.0, sr=44100), 1000.0, 0.707, sr=44100)[:, 0]
    assert np.allclose(y, [0, 0.000288428826, 0.0016714863, 0.00507197296, 0.0111980755], rtol=1e-6)
    st = O.blitsaw_state(0.0)
    y = O.blitsaw(st, 0, 5, 440.0, sr=44100)[:, 0]
    assert np.allclose(y, [0.616381526, 0.618029177, 0.386101067, 0.343552113, 0.449158311], rtol=1e-6)
    st = O.supersaw_state(7, seed=1234)
    assert np.allclose([float(s["initial_phase"][0]) for s in st["osc"]],
                       [0.97669977, 0.38019574, 0.92324623, 0.26169242, 0.31909706, 0.11809123,
                        0.24176629], atol=1e-8)
    y = O.supersaw(st, 0, 4, 440.0, sr=44100)[:, 0]
    assert np.allclose(y, [0.0634731874, 0.33049798, 0.568577588, 0.558507621], rtol=1e-6)
    st = O.ladder_state(1)
    y = O.ladder(st, np.ones((4, 1), np.float32), 1000.0, 0.0, sr=44100)[:, 0]
    assert np.allclose(y, [1.89822604e-05, 0.000176145943, 0.000719889358, 0.00197896571], rtol=1e-6)
    st = O.comb_state(2, 44100)
    y = O.comb(st, np.full((512, 2), 0.25, np.float32), 440.0, 0.7, sr=44100)
    assert y[0, 0] == 0.25 and np.isclose(y[100, 0], 0.425) and np.isclose(y[200, 1], 0.5475)
    assert np.isclose(y[511, 0], 0.735292494)
    st = O.convolve_state()
    y = O.convolve(st, 0, np.array([[1], [2], [3], [4], [0], [0]], np.float32),
                   np.array([1, 0.5, -1], np.float32), fft_size=16)[:, 0]
    assert np.allclose(y, [1, 2.5, 3, 3.5, -1, -4], atol=1e-6)


def test_convolve_matches_numpy_convolve():
    """tests/test_convolve_pe.py:49-162 of the reference: np.convolve equivalence, chunked == whole."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal(400).astype(np.float32)
    h = rng.standard_normal(33).astype(np.float32)
    want = np.convolve(x.astype(np.float64), h.astype(np.float64))[:400].astype(np.float32)
    st = O.convolve_state()
    whole = O.convolve(st, 0, x.reshape(-1, 1), h, fft_size=128)[:, 0]
    assert np.allclose(whole, want, atol=1e-5)
    st = O.convolve_state()
    pos, parts = 0, []
    for n in [17, 23, 19, 41, 7, 93, 200]:
        parts.append(O.convolve(st, pos, x[pos:pos + n].reshape(-1, 1), h, fft_size=128)[:, 0])
        pos += n
    assert np.allclose(np.concatenate(parts), want, atol=1e-5)
