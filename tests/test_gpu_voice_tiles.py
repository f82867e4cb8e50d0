"""GPU: the on-chip mix of BlitSaw -> Biquad [-> x envelope] voices (pgx_voice_tiles: no [voices][frames] layer between the
voices and the MixPE) against the layered path it replaces, the CPU oracle, and itself across seeks, resets, block lengths
and the two ways its tiles' entries are made (inline, or behind the block before)."""

import numpy as np
import pytest

import pygmu2_amd as pg
from oracle.golden_cases import S
from oracle.graph_eval import Node
from pygmu2_amd import voice_bank
from pygmu2_amd.sharding import c5_voice

pytestmark = pytest.mark.gpu

SR = 48000


def _render(root, blocks):
    r = pg.NullRenderer(sample_rate=SR)
    r.set_source(root)
    r.start()
    out = [root.render(s, n).data.copy() for s, n in blocks]
    r.stop()
    return out


@pytest.fixture
def small_banks_on_chip():
    keep = voice_bank.VOICE_TILES, voice_bank.VOICE_TILES_MIN_VOICES
    voice_bank.VOICE_TILES_MIN_VOICES = 4
    yield
    voice_bank.VOICE_TILES, voice_bank.VOICE_TILES_MIN_VOICES = keep


def _both(make, blocks):
    voice_bank.VOICE_TILES = True
    new = _render(make(), blocks)
    voice_bank.VOICE_TILES = False
    old = _render(make(), blocks)
    voice_bank.VOICE_TILES = True
    return new, old


# streamed blocks, a seek (blocks 3 -> 4), another length, a block that is not a multiple of 16 or 4
BLOCKS = [(0, 48000), (48000, 48000), (96000, 48000), (300000, 48000), (348000, 48000), (396000, 12345), (408345, 12345),
          (420690, 4099)]


@pytest.mark.parametrize("count", [5, 40, 200, 300])
def test_c5_voices_on_chip_mix_equals_layered_path(small_banks_on_chip, count):
    pg.set_sample_rate(SR)
    step = 512 // count
    new, old = _both(lambda: pg.MixPE(*[c5_voice(pg, i * step) for i in range(count)]), BLOCKS)
    peak = max(float(np.max(np.abs(b))) for b in old)
    for (s, n), a, b in zip(BLOCKS, new, old):
        assert a.shape == b.shape == (n, 1)
        assert np.max(np.abs(a - b)) <= 1e-6 * peak, f"block {(s, n)}: {np.max(np.abs(a - b)) / peak:.3e} of peak"


def test_stream_that_changes_paths_block_by_block(small_banks_on_chip):
    """Blocks below 4096 frames take the layered path, the others the on-chip mix: the states travel between them."""
    pg.set_sample_rate(SR)
    blocks, pos = [], 0
    for n in (48000, 1000, 48000, 2000, 4096, 4095, 30000, 30000, 512, 30000):
        blocks.append((pos, n))
        pos += n
    new, old = _both(lambda: pg.MixPE(*[c5_voice(pg, 11 * i) for i in range(40)]), blocks)
    peak = max(float(np.max(np.abs(b))) for b in old)
    for (s0, n), a, b in zip(blocks, new, old):
        assert np.max(np.abs(a - b)) <= 1e-6 * peak, f"block {(s0, n)}: {np.max(np.abs(a - b)) / peak:.3e} of peak"


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PGX_FUZZ_BANK_STREAMS", "16"))))
def test_random_pulls_with_everything_ahead_against_nothing_ahead(small_banks_on_chip, monkeypatch, seed):
    """test_gpu_bank_streams' pull patterns (streams with seeks, steps back, restarts, odd lengths) on banks that take the
    on-chip mix: envelopes one block ahead, the next block's entries behind the mix -- against every pull rendered when it
    is asked for.  (Found: a walk rendered on the main stream after a seek, followed by a walk ahead on a side stream that
    did not wait for it.)"""
    import test_gpu_bank_streams as streams
    rng = np.random.default_rng(81_000 + seed)
    count = int(rng.choice([6, 64, 130, 200, 260]))
    pulls = streams._pulls(rng)
    pg.set_sample_rate(SR)

    def run(ahead):
        for name in streams.SWITCHES:
            monkeypatch.setattr(voice_bank, name, ahead)
        return _render(pg.MixPE(*[c5_voice(pg, (3 * i) % 512) for i in range(count)]), pulls)

    got, want = run(True), run(False)
    peak = max(float(np.max(np.abs(w))) for w in want) or 1.0
    for i, ((s0, n), a, b) in enumerate(zip(pulls, got, want)):
        err = float(np.max(np.abs(a.astype(np.float64) - b)))
        assert err <= 1e-6 * peak, (count, i, pulls[max(0, i - 2):i + 1], err, peak)


def test_voices_without_gain_and_mixed_filters(small_banks_on_chip):
    pg.set_sample_rate(SR)

    def make():
        return pg.MixPE(*[pg.BiquadPE(pg.BlitSawPE(55.0 * 2 ** (i / 7.0), amplitude=0.3 + 0.01 * i),
                                      600.0 + 150.0 * i, 0.6 + 0.05 * (i % 5)) for i in range(24)])

    blocks = [(0, 20000), (20000, 20000), (40000, 20000), (7, 5000)]
    new, old = _both(make, blocks)
    peak = max(float(np.max(np.abs(b))) for b in old)
    for a, b in zip(new, old):
        assert np.max(np.abs(a - b)) <= 1e-6 * peak


def test_on_chip_mix_against_the_oracle(small_banks_on_chip):
    pg.set_sample_rate(SR)
    idx = [0, 37, 101, 256, 300, 411, 480, 511]
    blocks = [(0, 12000), (12000, 12000), (24000, 12000), (50000, 4096)]
    voice_bank.VOICE_TILES = True
    got = _render(pg.MixPE(*[c5_voice(pg, i) for i in idx]), blocks)
    spec = S("MixPE", inputs=[
        S("GainPE", source=S("BiquadPE", source=S("BlitSawPE", frequency=27.5 * 2 ** (i / 48.0)),
                             frequency=2000.0, q=0.707),
          gain=S("AdsrGatedPE", gate=S("PeriodicGate", frequency=2.0 + 0.01 * i, duty_cycle=0.5),
                 attack_time=0.01, decay_time=0.1, sustain_level=0.7, release_time=0.2)) for i in idx])
    oracle = Node(spec, SR)
    for (s, n), g in zip(blocks, got):
        w = oracle.render(s, n)
        assert np.max(np.abs(g.astype(np.float64) - w)) <= 1e-5 * np.max(np.abs(w)) + 1e-7


def test_path_is_taken_and_a_high_q_bank_keeps_the_layered_path(small_banks_on_chip):
    pg.set_sample_rate(SR)
    mix = pg.MixPE(*[c5_voice(pg, i) for i in range(8)])
    _render(mix, [(0, 8192)])
    source = mix._bank.root.children["source"]
    assert source.mixes_on_chip(8192) and source.rot_tables is not None and 0 < source.settle_fine <= 2048
    assert source.settle_fine % 16 == 0 and source.settle_fine <= source.settle
    assert not source.mixes_on_chip(4000)                     # short blocks: the layered path
    ringing = pg.MixPE(*[pg.BiquadPE(pg.BlitSawPE(110.0 + i), 500.0, 400.0) for i in range(8)])    # Q 400: thousands of frames
    _render(ringing, [(0, 8192)])
    assert not ringing._bank.root.mixes_on_chip(8192) and ringing._bank.root.rot_tables is None


def test_entries_made_ahead_are_the_entries_made_inline():
    """pgx_voice_tiles_entries for the next block of a stream (advance = n, from this block's start states) writes, bit for
    bit, what pgx_voice_tiles would make for itself once the states have moved on."""
    from pygmu2_amd import device
    from pygmu2_amd._kernels import DeviceBuffer, check, lib

    pg.set_sample_rate(SR)
    L = lib()
    k, n = 12, 20000
    voice_bank_min = voice_bank.VOICE_TILES_MIN_VOICES
    voice_bank.VOICE_TILES_MIN_VOICES = 4
    try:
        mix = pg.MixPE(*[pg.BiquadPE(pg.BlitSawPE(40.0 * 2 ** (i / 5.0)), 1500.0, 0.8) for i in range(k)])
        first = _render(mix, [(0, n)])[0]
        node = mix._bank.root
        src = node.children["source"]
        warm = node.settle_fine
        ws_bytes = L.pgx_voice_tiles_workspace_bytes(k, n, warm)
        a = DeviceBuffer((ws_bytes,), np.uint8, zero=True)
        b = DeviceBuffer((ws_bytes,), np.uint8, zero=True)
        # `src.state_alt` holds the states block 0 started from, `src.state` the ones it ended with
        check(L.pgx_voice_tiles_entries(a.ptr, 1, k, n, node.rot_tables.ptr, src.state_alt.ptr, n, warm), "entries ahead")
        check(L.pgx_voice_tiles_entries(b.ptr, 1, k, n, node.rot_tables.ptr, src.state.ptr, 0, warm), "entries inline")
        device.synchronize()
        ea, eb = a.to_host(), b.to_host()
        assert np.array_equal(ea, eb) and np.any(ea != 0)
        assert first.shape == (n, 1)
    finally:
        voice_bank.VOICE_TILES_MIN_VOICES = voice_bank_min


def test_full_bank_streams_like_the_layered_path():
    """512 voices, the bench's block length: the default path of C5 (no fixture: the thresholds as shipped)."""
    pg.set_sample_rate(SR)
    blocks = [(i * 48000, 48000) for i in range(4)]
    new, old = _both(lambda: pg.MixPE(*[c5_voice(pg, i) for i in range(512)]), blocks)
    peak = max(float(np.max(np.abs(b))) for b in old)
    for a, b in zip(new, old):
        assert np.max(np.abs(a - b)) <= 1.2e-6 * peak          # (512 float32 additions in voice order on the layered side)
