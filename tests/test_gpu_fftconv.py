"""
GPU: the FFT overlap-save path of ConvolvePE (pgx_convolve_fft) against numpy's float64
convolution and against the direct MFMA path, over geometries that exercise every branch:
square and non-square N1 x N2, odd and even numbers of overlap-save blocks (packed pairs),
blocks shorter than the filter, carried history, every channel rule.
"""

import numpy as np
import pytest
from scipy.signal import fftconvolve

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    from pygmu2_amd import device
    device.ensure_init()
    return device


def _fft_conv(dev, x, h, hist, fft_size, hist_is_zero=0):
    lib = dev.ensure_init()
    n, src_ch = x.shape
    L, fir_ch = h.shape
    out_ch = max(src_ch, fir_ch)
    hd = dev.DeviceBuffer.from_host(h)
    spec = dev.DeviceBuffer((lib.pgx_convolve_fft_spectrum_bytes(fft_size, fir_ch),), np.uint8)
    dev.check(lib.pgx_convolve_fft_prepare(spec.ptr, hd.ptr, L, fir_ch, fft_size))
    xd = dev.DeviceBuffer.from_host(x)
    histd = dev.DeviceBuffer.from_host(hist)
    ws = dev.DeviceBuffer((lib.pgx_convolve_fft_workspace_bytes(n, L, out_ch, fft_size),), np.uint8)
    out = dev.DeviceBuffer((n, out_ch), np.float32)
    dev.check(lib.pgx_convolve_fft(out.ptr, xd.ptr, n, src_ch, spec.ptr, L, fir_ch, out_ch, fft_size, histd.ptr,
                                   ws.ptr, hist_is_zero))
    return out.to_host(), histd.to_host()


def _numpy_conv(x, h, hist):
    n, src_ch = x.shape
    L, fir_ch = h.shape
    out_ch = max(src_ch, fir_ch)
    y = np.zeros((n, out_ch))
    new_hist = np.zeros_like(hist)
    for c in range(out_ch):
        xc = x[:, 0 if src_ch == 1 else c].astype(np.float64)
        hc = h[:, 0 if fir_ch == 1 else c].astype(np.float64)
        ext = np.concatenate([hist[:, c].astype(np.float64), xc])
        y[:, c] = fftconvolve(ext, hc)[L - 1:L - 1 + n]          # float64, ~1e-13: a reference, not the oracle
        new_hist[:, c] = ext[n:n + L - 1]
    return y, new_hist


@pytest.mark.parametrize("L,n,fft_size", [
    (300, 2500, 4096),          # N1 = N2 = 64, one block
    (2000, 9000, 4096),         # 5 blocks: odd count -> last pair half empty
    (3000, 1000, 8192),         # block shorter than the filter, N1 = 64, N2 = 128
    (5000, 40000, 16384),       # 128 x 128
    (20000, 70000, 65536),      # 256 x 256, two blocks
    (65536, 96000, 131072),     # C3: 256 x 512
    (100000, 30000, 262144),    # 512 x 512
])
@pytest.mark.parametrize("src_ch,fir_ch", [(1, 1), (2, 1), (2, 2), (1, 2)])
def test_fft_conv_matches_numpy(dev, L, n, fft_size, src_ch, fir_ch):
    if L >= 65536 and (src_ch, fir_ch) not in ((2, 1), (2, 2)):
        pytest.skip("big cases: two channel layouts are enough")
    rng = np.random.default_rng(L + n)
    out_ch = max(src_ch, fir_ch)
    x = (rng.standard_normal((n, src_ch)) * 0.1).astype(np.float32)
    h = (rng.standard_normal((L, fir_ch)) * np.exp(-np.arange(L) / (L / 8.0))[:, None]).astype(np.float32)
    hist = (rng.standard_normal((L - 1, out_ch)) * 0.1).astype(np.float32)
    got, got_hist = _fft_conv(dev, x, h, hist, fft_size)
    want, want_hist = _numpy_conv(x, h, hist)
    peak = float(np.max(np.abs(want)))
    err = float(np.max(np.abs(got.astype(np.float64) - want)))
    assert err <= 1e-6 * peak, (err, peak)              # float64 transforms: float32 rounding only
    assert np.array_equal(got_hist, want_hist.astype(np.float32))
    # a fresh stream: whatever the history buffer holds counts as zeros, and it is still rewritten
    got0, got_hist0 = _fft_conv(dev, x, h, hist, fft_size, hist_is_zero=1)
    want0, want_hist0 = _numpy_conv(x, h, np.zeros_like(hist))
    assert float(np.max(np.abs(got0.astype(np.float64) - want0))) <= 1e-6 * float(np.max(np.abs(want0)))
    assert np.array_equal(got_hist0, want_hist0.astype(np.float32))


def test_fft_size_rule_and_argument_checks(dev):
    lib = dev.ensure_init()
    assert lib.pgx_convolve_fft_size(1) == 4096
    assert lib.pgx_convolve_fft_size(2049) == 8192
    assert lib.pgx_convolve_fft_size(65536) == 131072
    assert lib.pgx_convolve_fft_size(131072) == 262144
    assert lib.pgx_convolve_fft_size(131073) == 0                      # too long: the caller stays on pgx_convolve
    assert lib.pgx_convolve_fft_workspace_bytes(1000, 300, 2, 3000) == 0     # not a power of two
    assert lib.pgx_convolve_fft_workspace_bytes(1000, 5000, 2, 4096) == 0    # hop would be < 1


def test_convolve_pe_streams_through_the_fft_path(dev):
    """ConvolvePE with a 20 000-tap filter in uneven blocks == one numpy convolution."""
    import pygmu2_amd as pg
    from pygmu2_amd import convolve_pe
    pg.set_sample_rate(48000)
    rng = np.random.default_rng(8)
    L, T = 20000, 150_000
    assert L >= convolve_pe.FFT_MIN_TAPS
    x = (rng.standard_normal((T, 2)) * 0.1).astype(np.float32)
    h = (rng.standard_normal(L) * np.exp(-np.arange(L) / 3000.0)).astype(np.float32)
    pe = pg.ConvolvePE(pg.ArrayPE(x), pg.ArrayPE(h))
    r = pg.NullRenderer(sample_rate=48000)
    r.set_source(pe)
    r.start()
    sizes = [48000, 17, 30000, 1, 71982]
    pos, parts = 0, []
    for n in sizes:
        parts.append(pe.render(pos, n).data)
        pos += n
    r.stop()
    assert pe._device_fft == 65536
    got = np.concatenate(parts)
    want = np.stack([fftconvolve(x[:, c].astype(np.float64), h.astype(np.float64))[:T] for c in range(2)], axis=1)
    peak = float(np.max(np.abs(want)))
    assert float(np.max(np.abs(got.astype(np.float64) - want))) <= REL_TOL * peak


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PGX_FUZZ_FFTCONV", "40"))))
def test_fft_conv_random_geometries(dev, seed):
    """Random filter lengths (2 .. 100 000 taps), block lengths (1 .. 200 000 frames, also just around multiples of the
    hop), any transform size the library accepts for the filter (the smallest and larger ones), every channel rule, a
    carried history or a fresh stream -- against numpy's float64 convolution."""
    lib = dev.ensure_init()
    rng = np.random.default_rng(31_000 + seed)
    L = int(rng.choice([2, 3, 63, 64, 65, 2047, 2048, 2049, int(rng.integers(3, 100_000))]))      # (fir_len >= 2: the ABI's rule)
    smallest = lib.pgx_convolve_fft_size(L)
    assert smallest
    sizes = [f for f in (4096, 8192, 16384, 32768, 65536, 131072, 262144) if f >= smallest]
    fft_size = int(sizes[int(rng.integers(0, min(3, len(sizes))))])
    hop = fft_size - (L - 1)
    n = int(rng.choice([1, hop - 1, hop, hop + 1, 2 * hop, 3 * hop + 1, int(rng.integers(1, 200_000))]))
    n = max(1, min(n, 400_000))
    src_ch, fir_ch = [(1, 1), (2, 1), (2, 2), (1, 2)][int(rng.integers(0, 4))]
    out_ch = max(src_ch, fir_ch)
    x = (rng.standard_normal((n, src_ch)) * 0.1).astype(np.float32)
    h = (rng.standard_normal((L, fir_ch)) * np.exp(-np.arange(L) / max(1.0, L / 8.0))[:, None]).astype(np.float32)
    fresh = int(rng.random() < 0.3)
    hist = (rng.standard_normal((max(L - 1, 0), out_ch)) * 0.1).astype(np.float32)
    got, got_hist = _fft_conv(dev, x, h, hist, fft_size, hist_is_zero=fresh)
    want, want_hist = _numpy_conv(x, h, np.zeros_like(hist) if fresh else hist)
    assert np.array_equal(got_hist, want_hist.astype(np.float32)), (L, n, fft_size, src_ch, fir_ch, fresh)
    peak = float(np.max(np.abs(want))) or 1.0
    err = float(np.max(np.abs(got.astype(np.float64) - want)))
    assert err <= 1e-6 * peak + 1e-9, (err, peak, L, n, fft_size, src_ch, fir_ch, fresh)
