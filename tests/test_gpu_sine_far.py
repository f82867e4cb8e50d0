"""GPU: SinePE far from the stream origin.  The sample time is n / sr rounded as the division rounds it and the
sine argument is reduced by the fused Cody-Waite steps up to 2e9 rad (pgx_common.h); beyond that the library
routine takes over.  Each window against the oracle (numpy float64 sine, sine_pe.py:119-175)."""

import numpy as np
import pytest

import pygmu2_amd as pg
from oracle import pe_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sr", [44100, 48000])
@pytest.mark.parametrize("start", [0, 10 ** 7, 10 ** 9 + 7, 3 * 10 ** 9, 2 * 10 ** 10 + 12345, 10 ** 11, 2 ** 40 + 3,
                                   -(10 ** 9) - 5])
def test_sine_far_windows_match_numpy(start, sr):
    pg.set_sample_rate(sr)
    for channels, n in ((1, 8192), (1, 4099), (2, 1000)):          # mono fast path, ragged tail, frame replication
        pe = pg.SinePE(frequency=440.0, amplitude=0.9, phase=0.25, channels=channels)
        got = pe.render(start, n).data
        want = pe_oracle.sine_pure(start, n, 440.0, 0.9, 0.25, sr, channels)
        diff = got != want
        # the float64 sines agree to an ulp or two: a float32 result may fall on the other side of a rounding
        # boundary once in ~1e7 samples, never by more than one float32 ulp
        assert np.count_nonzero(diff) <= 1, (start, sr, channels, int(np.count_nonzero(diff)))
        assert np.max(np.abs(got.astype(np.float64) - want)) <= 6.0e-8
