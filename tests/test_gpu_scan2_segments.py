"""
GPU: the segmented (reduce + apply) form of the time-varying 2x2 scans -- BiquadPE with PE-driven
frequency / Q and SVFilterPE -- on blocks long enough to be cut into many segments, against the oracle
(the C restatement of the reference's numba kernels), with state streamed across blocks of awkward sizes.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5
SIZES = [44100, 1025, 3000, 1, 2048, 100_000, 44100]        # 1 .. 98 tiles, single- and multi-segment plans


def _stream(pe, pg, sr):
    r = pg.NullRenderer(sample_rate=sr)
    r.set_source(pe)
    r.start()
    pos, parts = 0, []
    for n in SIZES:
        parts.append(pe.render(pos, n).data)
        pos += n
    r.stop()
    return np.concatenate(parts)


def _oracle_stream(spec, sr):
    from oracle import graph_eval
    g = graph_eval.Node(spec, sr)
    pos, parts = 0, []
    for n in SIZES:
        parts.append(g.render(pos, n))
        pos += n
    return np.concatenate(parts)


@pytest.mark.parametrize("kind,mode,var", [
    ("BiquadPE", "lowpass", "frequency"), ("BiquadPE", "bandpass", "q"), ("BiquadPE", "peaking", "both"),
    ("SVFilterPE", "lowpass", "frequency"), ("SVFilterPE", "highshelf", "both"), ("SVFilterPE", "bandpass", "const"),
])
def test_segmented_scan_matches_oracle(kind, mode, var):
    import pygmu2_amd as pg
    from oracle.golden_cases import S
    import spec_build
    sr = 44100
    pg.set_sample_rate(sr)
    total = sum(SIZES)
    noise = {"rng": 21, "n": total, "ch": 2, "scale": 0.3}
    f = S("MixPE", inputs=[S("ConstantPE", value=1200.0), S("SinePE", frequency=3.0, amplitude=700.0)])
    q = S("MixPE", inputs=[S("ConstantPE", value=2.0), S("SinePE", frequency=1.3, amplitude=1.2)])
    spec = S(kind, source=S("ArrayPE", data=noise), mode=mode, gain_db=5.0,
             frequency=f if var in ("frequency", "both") else 1000.0, q=q if var in ("q", "both") else 1.7)
    got = _stream(spec_build.build(spec), pg, sr)
    want = _oracle_stream(spec, sr)
    assert got.shape == want.shape == (total, 2)
    peak = float(np.max(np.abs(want)))
    err = float(np.max(np.abs(got.astype(np.float64) - want)))
    assert err <= REL_TOL * peak, (kind, mode, var, err, peak)
