"""
GPU: the single-launch ("settled") path of pgx_biquad_const against the exact reduce + apply
pair and against the oracle (scipy.signal.lfilter restatement of biquad_pe.py:383-404).

The settled path may only be chosen by a host that has bounded A^W (biquad_pe.settle_frames);
these tests pin that its output is the exact path's to far below the parity tolerance, for
aligned and unaligned buffers, mono and stereo, carried state and every block length class.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5        # north_star tolerance, relative to peak


@pytest.fixture(scope="module")
def env():
    import pygmu2_amd as pg
    from pygmu2_amd import device
    from pygmu2_amd.biquad_pe import rbj_coefficients, settle_frames
    lib = device.ensure_init()

    class Env:
        pass

    e = Env()
    e.pg, e.device, e.lib = pg, device, lib
    e.rbj, e.settle = rbj_coefficients, settle_frames
    return e


def _run(e, x, coef, settle, state0=None, offset=0):
    """One pgx_biquad_const call on x (frames, channels); returns (y, final state)."""
    device, lib = e.device, e.lib
    n, ch = x.shape
    # `offset` floats of slack in front of both buffers -> unaligned base pointers
    xin = device.DeviceBuffer((n * ch + offset,), np.float32)
    xin.upload(np.concatenate([np.zeros(offset, np.float32), x.reshape(-1)]))
    out = device.DeviceBuffer((n * ch + offset,), np.float32)
    cbuf = device.DeviceBuffer.from_host(np.asarray(coef, dtype=np.float64))
    tables = device.DeviceBuffer((lib.pgx_biquad_table_doubles(),), np.float64)
    device.check(lib.pgx_biquad_tables(tables.ptr, cbuf.ptr, 1))
    st = device.DeviceBuffer.from_host(np.zeros((ch, 2)) if state0 is None else np.asarray(state0, dtype=np.float64))
    need = lib.pgx_biquad_workspace_bytes(1, n, ch, settle)
    ws = device.DeviceBuffer((max(need, 1),), np.uint8)
    device.check(lib.pgx_biquad_const(out.offset_ptr(offset), 0, xin.offset_ptr(offset), 0, 1, n, ch, cbuf.ptr,
                                      tables.ptr if settle else None, settle, st.ptr, ws.ptr))
    return out.to_host()[offset:].reshape(n, ch), st.to_host()


@pytest.mark.parametrize("n", [8191, 8193, 12288, 40_000, 44_100, 100_003, 1_000_000, 3_333_333])
@pytest.mark.parametrize("ch,offset", [(1, 0), (1, 1), (2, 0)])
def test_settled_equals_exact(env, n, ch, offset):
    rng = np.random.default_rng(n + ch)
    x = (rng.standard_normal((n, ch)) * 0.3).astype(np.float32)
    coef = env.rbj(env.pg.BiquadMode.LOWPASS, 1000.0, 0.707, 0.0, 44100.0)
    w = env.settle(coef[3], coef[4])
    assert 0 < w <= 4096
    s0 = rng.standard_normal((ch, 2)) * 0.1
    y_exact, st_exact = _run(env, x, coef, 0, s0, offset)
    y_settled, st_settled = _run(env, x, coef, w, s0, offset)
    peak = float(np.max(np.abs(y_exact)))
    err = float(np.max(np.abs(y_settled.astype(np.float64) - y_exact)))
    assert err <= 1e-7 * peak, (n, ch, offset, err, peak)
    assert np.allclose(st_settled, st_exact, rtol=1e-12, atol=1e-15 * peak)


@pytest.mark.parametrize("mode,freq,q,gain", [
    ("lowpass", 1000.0, 0.707, 0.0), ("highpass", 200.0, 1.5, 0.0), ("bandpass", 3000.0, 4.0, 0.0),
    ("peaking", 800.0, 2.0, 9.0), ("lowshelf", 150.0, 0.8, -6.0), ("notch", 60.0, 5.0, 0.0)])
def test_settled_matches_oracle(env, mode, freq, q, gain):
    from oracle import pe_oracle as O
    n = 300_000
    x = (np.random.default_rng(5).standard_normal((n, 1)) * 0.25).astype(np.float32)
    coef = env.rbj(env.pg.BiquadMode(mode), freq, q, gain, 48000.0)
    w = env.settle(coef[3], coef[4])
    st = O.biquad_state(1)
    want = O.biquad_const(st, x, freq, q, mode=mode, gain_db=gain, sr=48000)
    got, _ = _run(env, x, coef, w)
    peak = float(np.max(np.abs(want)))
    assert float(np.max(np.abs(got.astype(np.float64) - want))) <= REL_TOL * peak + 1e-7


def test_slow_decay_falls_back_to_exact_pair(env):
    coef = env.rbj(env.pg.BiquadMode.LOWPASS, 20.0, 10.0, 0.0, 48000.0)
    assert env.settle(coef[3], coef[4]) == 0                  # |pole| ~ 0.99987: no usable W
    assert env.lib.pgx_biquad_workspace_bytes(1, 1_000_000, 1, 0) > 0
    coef = env.rbj(env.pg.BiquadMode.LOWPASS, 100.0, 0.707, 0.0, 48000.0)
    w = env.settle(coef[3], coef[4])
    assert w == 8192                                           # two halves of warm-up
    x = (np.random.default_rng(9).standard_normal((500_000, 1)) * 0.3).astype(np.float32)
    y_exact, _ = _run(env, x, coef, 0)
    y_settled, _ = _run(env, x, coef, w)
    peak = float(np.max(np.abs(y_exact)))
    assert float(np.max(np.abs(y_settled.astype(np.float64) - y_exact))) <= 1e-7 * peak


def test_biquad_pe_streams_state_through_settled_blocks(env):
    """BiquadPE over 3 x 400k-frame blocks (settled path each) == one oracle pass over 1.2M frames."""
    from oracle import pe_oracle as O
    pg = env.pg
    pg.set_sample_rate(44100)
    pe = pg.BiquadPE(pg.SinePE(440.0), frequency=1000.0, q=0.707)
    r = pg.NullRenderer(sample_rate=44100)
    r.set_source(pe)
    r.start()
    got = np.concatenate([pe.render(i * 400_000, 400_000).data for i in range(3)])
    r.stop()
    assert pe._settle > 0 and pe._tables is not None
    st = O.biquad_state(1)
    want = O.biquad_const(st, O.sine_pure(0, 1_200_000, 440.0, sr=44100), 1000.0, 0.707, sr=44100)
    peak = float(np.max(np.abs(want)))
    assert float(np.max(np.abs(got.astype(np.float64) - want))) <= REL_TOL * peak
