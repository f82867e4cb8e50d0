"""GPU: batched voice banks and the sharded mix produce the same samples as rendering the
voices one by one, and stay within the parity budget of the CPU oracle."""

import numpy as np
import pytest

import pygmu2_amd as pg
from oracle.golden_cases import S
from oracle.graph_eval import Node
from pygmu2_amd.sharding import ShardedMixPE, c5_voice
from spec_build import build

pytestmark = pytest.mark.gpu


def _render_blocks(root, sr, blocks):
    r = pg.NullRenderer(sample_rate=sr)
    r.set_source(root)
    r.start()
    out = [root.render(s, n).data for s, n in blocks]
    r.stop()
    return out


def test_c5_bank_equals_unbatched_and_oracle():
    pg.set_sample_rate(48000)
    idx = [0, 37, 101, 256, 300, 411, 480, 511]
    blocks = [(0, 12000), (12000, 12000), (24000, 1000), (50000, 4096)]     # last one: a gap -> resets

    banked = pg.MixPE(*[c5_voice(pg, i) for i in idx])
    got_bank = _render_blocks(banked, 48000, blocks)
    assert banked._bank, "voice bank was not built for identical C5 voices"

    plain = pg.MixPE(*[c5_voice(pg, i) for i in idx])
    plain._bank = False                                                     # force per-input rendering
    got_plain = _render_blocks(plain, 48000, blocks)
    # (both paths run the sixteen-frames-per-thread oscillator, cut into time segments differently: the same samples to
    # a float32 ulp here and there -- they happened to be bit-identical until the denominators' recurrence changed at the
    # end of round 4; `voice_bank.WIDE_SUPERSAW = False` selects the kernels that are bit-identical by construction)
    peak = max(float(np.max(np.abs(b))) for b in got_plain)
    for a, b in zip(got_bank, got_plain):
        assert np.max(np.abs(a - b)) <= 2e-7 * peak, f"bank differs from per-voice path, max|d|={np.max(np.abs(a - b))}"

    # ... and with the eight-frames-per-thread kernels on both sides, bit for bit
    from pygmu2_amd import blit_saw_pe, voice_bank
    keep = voice_bank.WIDE_SUPERSAW, blit_saw_pe.WIDE_LONG_RENDERS
    voice_bank.WIDE_SUPERSAW = blit_saw_pe.WIDE_LONG_RENDERS = False
    try:
        exact_bank = _render_blocks(pg.MixPE(*[c5_voice(pg, i) for i in idx]), 48000, blocks)
        exact_plain_mix = pg.MixPE(*[c5_voice(pg, i) for i in idx])
        exact_plain_mix._bank = False
        exact_plain = _render_blocks(exact_plain_mix, 48000, blocks)
    finally:
        voice_bank.WIDE_SUPERSAW, blit_saw_pe.WIDE_LONG_RENDERS = keep
    for a, b in zip(exact_bank, exact_plain):
        assert np.array_equal(a, b), f"exact bank differs from per-voice path, max|d|={np.max(np.abs(a - b))}"

    spec = S("MixPE", inputs=[
        S("GainPE", source=S("BiquadPE", source=S("BlitSawPE", frequency=27.5 * 2 ** (i / 48.0)),
                             frequency=2000.0, q=0.707),
          gain=S("AdsrGatedPE", gate=S("PeriodicGate", frequency=2.0 + 0.01 * i, duty_cycle=0.5),
                 attack_time=0.01, decay_time=0.1, sustain_level=0.7, release_time=0.2)) for i in idx])
    oracle = Node(spec, 48000)
    for (s, n), g in zip(blocks, got_bank):
        w = oracle.render(s, n)
        assert np.max(np.abs(g.astype(np.float64) - w)) <= 1e-5 * np.max(np.abs(w)) + 1e-7


def test_c4_bank_supersaw_ladder():
    pg.set_sample_rate(48000)

    def voice(i):
        return pg.LadderPE(pg.SuperSawPE(55.0 * 2 ** (i / 12.0), voices=7, detune_cents=20.0, seed=i),
                           frequency=1200.0, resonance=0.3, mode=pg.LadderMode.LP24, drive=1.0, oversample=2)

    # from 4096 frames on the bank renders the next block's oscillators beside this block's ladder: the third
    # block is a seek and the fifth another length, so that speculation is rolled back twice
    blocks = [(0, 4096), (4096, 4096), (30000, 4096), (34096, 4096), (38192, 5000), (43192, 5000)]
    banked = pg.MixPE(*[voice(i) for i in range(6)])
    got_bank = _render_blocks(banked, 48000, blocks)
    assert banked._bank
    again = _render_blocks(banked, 48000, blocks)               # stop/start in between: the same samples again
    for a, b in zip(got_bank, again):
        assert np.array_equal(a, b)
    plain = pg.MixPE(*[voice(i) for i in range(6)])
    plain._bank = False
    got_plain = _render_blocks(plain, 48000, blocks)
    for a, b in zip(got_bank, got_plain):
        # the bank's oscillators run in concurrent time segments (closed-form integrator carries, ~1e-14 in float64)
        assert float(np.max(np.abs(a.astype(np.float64) - b))) <= 1e-6 * float(np.max(np.abs(b)))
    spec = S("MixPE", inputs=[
        S("LadderPE", source=S("SuperSawPE", frequency=55.0 * 2 ** (i / 12.0), voices=7, detune_cents=20.0, seed=i),
          frequency=1200.0, resonance=0.3, mode="lp24", drive=1.0, oversample=2) for i in range(6)])
    oracle = Node(spec, 48000)
    for (s, n), g in zip(blocks, got_bank):
        w = oracle.render(s, n)
        assert np.max(np.abs(g.astype(np.float64) - w)) <= 1e-5 * np.max(np.abs(w)) + 1e-7


def test_bank_restart_reproduces_first_run():
    pg.set_sample_rate(48000)
    root = pg.MixPE(*[c5_voice(pg, i) for i in range(5)])
    first = _render_blocks(root, 48000, [(0, 8000), (8000, 8000)])
    again = _render_blocks(root, 48000, [(0, 8000), (8000, 8000)])
    for a, b in zip(first, again):
        assert np.array_equal(a, b)


def test_not_batchable_graphs_fall_back():
    pg.set_sample_rate(48000)
    mixed = pg.MixPE(pg.SinePE(440.0), pg.BlitSawPE(220.0), pg.SinePE(330.0), pg.SinePE(550.0))
    mixed.render(0, 256)
    assert mixed._bank is False
    shared = pg.SinePE(100.0)
    m2 = pg.MixPE(shared, shared, shared, shared)               # one node reused: not private voices
    m2.render(0, 256)
    assert m2._bank is False
    cropped = pg.MixPE(*[pg.CropPE(pg.SinePE(100.0 * (i + 1)), 0, 100) for i in range(4)])
    cropped.render(0, 256)
    assert cropped._bank is False


def test_sharded_world1_equals_mix():
    pg.set_sample_rate(48000)
    a = ShardedMixPE([c5_voice(pg, i) for i in range(6)], 0, 1)
    b = pg.MixPE(*[c5_voice(pg, i) for i in range(6)])
    ga = _render_blocks(a, 48000, [(0, 6000), (6000, 6000)])
    gb = _render_blocks(b, 48000, [(0, 6000), (6000, 6000)])
    for x, y in zip(ga, gb):
        assert np.array_equal(x, y)


def test_wide_blitsaw_workgroups_reproduce_the_bank_kernel_bit_for_bit():
    """A lone oscillator renders with 512-thread workgroups (4096-frame tiles), an oscillator inside a bank of
    128+ with 256-thread ones (2048-frame tiles): the wave values are folded in the same order, so the samples
    must be identical -- also across block boundaries and with a partial last tile."""
    import numpy as np
    from pygmu2_amd import device
    lib = device.ensure_init()
    sr = 48000.0
    freqs = [27.5 * 2 ** (i / 48.0) for i in range(130)]
    rec = np.zeros(len(freqs), dtype=device.BLITSAW_PARAMS)
    for i, f in enumerate(freqs):
        rec[i] = (f, 1.0, 0.999, 0.0)
    params = device.upload_structs(rec)
    blocks = [20000, 8192, 8193, 3000, 100000, 12289]
    state_bank = device.DeviceBuffer((len(freqs), 2), np.float64, zero=True)
    state_one = device.DeviceBuffer((1, 2), np.float64, zero=True)
    state_seg = device.DeviceBuffer((3, 2), np.float64, zero=True)
    pick = 77
    one_params = device.upload_structs(rec[pick:pick + 1])
    seg_params = device.upload_structs(rec[pick - 1:pick + 2])
    for n in blocks:
        bank = device.DeviceBuffer((len(freqs), n, 1), np.float32)
        device.check(lib.pgx_blitsaw(bank.ptr, n, len(freqs), n, 1, sr, params.ptr, None, 0, None, 0, None, 0,
                                     state_bank.ptr, None, None))
        one = device.DeviceBuffer((1, n, 1), np.float32)
        device.check(lib.pgx_blitsaw(one.ptr, n, 1, n, 1, sr, one_params.ptr, None, 0, None, 0, None, 0,
                                     state_one.ptr, None, None))
        assert np.array_equal(bank.to_host()[pick], one.to_host()[0]), n
        # several workgroups per oscillator (two passes over a workspace): the same chains replayed
        need = lib.pgx_blitsaw_workspace_bytes(3, n, 0)
        assert (need > 0) == (n > 2 * 4096), (n, need)
        ws = device.DeviceBuffer((max(need, 8),), np.uint8)
        seg = device.DeviceBuffer((3, n, 1), np.float32)
        device.check(lib.pgx_blitsaw(seg.ptr, n, 3, n, 1, sr, seg_params.ptr, None, 0, None, 0, None, 0,
                                     state_seg.ptr, ws.ptr if need else None, None))
        assert np.array_equal(seg.to_host()[1], one.to_host()[0]), n
        assert np.array_equal(seg.to_host()[0], bank.to_host()[pick - 1]), n
        assert np.array_equal(seg.to_host()[2], bank.to_host()[pick + 1]), n
    assert np.array_equal(state_seg.to_host()[1], state_one.to_host()[0])
    assert np.array_equal(state_bank.to_host()[pick], state_one.to_host()[0])


def test_blitsaw_long_stream_in_segments_is_the_single_workgroup_stream():
    """A 3 M-frame stream of a few oscillators (what a look-ahead window of 64 x 44 100-frame blocks hands the
    kernel): 733 tiles in 245 segments, the integrator chain folded by k_blitsaw_chain over 92 chunks of 64 wave
    responses -- against one workgroup per oscillator walking all tiles, bit for bit, states included, with a
    second block carried across and the state snapshot (state_backup) of the segmented form checked."""
    from pygmu2_amd import device
    lib = device.ensure_init()
    sr = 44100.0
    rec = np.zeros(3, dtype=device.BLITSAW_PARAMS)
    for i, f in enumerate((440.0, 61.7, 3520.0)):
        rec[i] = (f, 0.8, 0.999, 0.0)
    params = device.upload_structs(rec)
    st_seg = device.DeviceBuffer((3, 2), np.float64, zero=True)
    st_one = device.DeviceBuffer((3, 2), np.float64, zero=True)
    for n in (3_000_000, 70_001):
        need = lib.pgx_blitsaw_workspace_bytes(3, n, 0)
        assert need > 0
        ws = device.DeviceBuffer((need,), np.uint8)
        before = st_seg.to_host().copy()
        backup = device.DeviceBuffer((3, 2), np.float64, zero=True)
        seg = device.DeviceBuffer((3, n, 1), np.float32)
        device.check(lib.pgx_blitsaw(seg.ptr, n, 3, n, 1, sr, params.ptr, None, 0, None, 0, None, 0, st_seg.ptr,
                                     ws.ptr, backup.ptr))
        one = device.DeviceBuffer((3, n, 1), np.float32)
        device.check(lib.pgx_blitsaw(one.ptr, n, 3, n, 1, sr, params.ptr, None, 0, None, 0, None, 0, st_one.ptr,
                                     None, None))
        assert np.array_equal(seg.to_host(), one.to_host()), n
        assert np.array_equal(st_seg.to_host(), st_one.to_host()), n
        assert np.array_equal(backup.to_host(), before), n


@pytest.mark.parametrize("voices,channels", [(7, 1), (3, 2), (1, 1), (16, 1)])
def test_supersaw_bank_summed_on_chip_is_the_two_launch_path_bit_for_bit(voices, channels, monkeypatch):
    """pgx_supersaw_bank (voices accumulated inside the oscillator kernel, no intermediate) against
    pgx_blitsaw + pgx_supersaw_sum and against per-voice rendering: identical float32 samples, identical
    carried state (a second and a third block, a gap that resets, an unaligned tail)."""
    from pygmu2_amd import voice_bank
    pg.set_sample_rate(48000)
    monkeypatch.setattr(voice_bank, "WIDE_SUPERSAW", False)           # the 8-frames-per-thread bank: k_blitsaw's bits
    from pygmu2_amd import blit_saw_pe
    monkeypatch.setattr(blit_saw_pe, "WIDE_LONG_RENDERS", False)      # (and for the lone PEs inside look-ahead windows)
    n_inst = 258 if voices == 7 else 256                              # (pgx_supersaw_wide: test_gpu_supersaw_segments.py)

    def make():
        return pg.MixPE(*[pg.SuperSawPE(55.0 * 2 ** (i / 24.0), amplitude=0.5 + 0.001 * i, voices=voices,
                                        detune_cents=20.0, seed=i, channels=channels,
                                        mix_mode=("center_heavy", "linear", "equal")[i % 3]) for i in range(n_inst)])

    blocks = [(0, 5000), (5000, 4096), (9096, 8191), (40000, 777)]
    fused = make()
    got_fused = _render_blocks(fused, 48000, blocks)
    assert fused._bank and fused._bank.root.k == n_inst
    monkeypatch.setattr(voice_bank, "FUSED_SUPERSAW_MIN", 10 ** 9)
    monkeypatch.setattr(voice_bank, "SEGMENTED_SUPERSAW", False)      # (tests/test_gpu_supersaw_segments.py)
    two = make()
    got_two = _render_blocks(two, 48000, blocks)
    for a, b in zip(got_fused, got_two):
        assert a.shape == b.shape and np.array_equal(a, b), float(np.max(np.abs(a - b)))
    plain = make()
    plain._bank = False
    got_plain = _render_blocks(plain, 48000, blocks[:2])
    for a, b in zip(got_fused, got_plain):
        assert np.array_equal(a, b), float(np.max(np.abs(a - b)))


@pytest.mark.parametrize("wide", [False, True])
def test_blitsaw_biquad_bank_in_one_launch_matches_the_two_launch_bank(monkeypatch, wide):
    """pgx_blitsaw_biquad_bank (oscillator samples filtered in registers) against pgx_blitsaw + pgx_biquad_const
    over the same 140 voices: the oscillator bits are the same, the filter's carry-ins come out of a differently
    shaped scan (so a float32 sample may differ in its last bit once in a long while), states carry over blocks,
    a gap resets the oscillators.  wide: pgx_blitsaw_biquad_wide (sixteen frames per thread, pgx_supersaw_wide's
    oscillator: phases as products, the numerator by a three-term recurrence) -- <= 1e-6 of peak."""
    from pygmu2_amd import voice_bank
    pg.set_sample_rate(48000)
    monkeypatch.setattr(voice_bank, "WIDE_SUPERSAW", wide)
    idx = list(range(0, 512, 4)) + list(range(1, 48, 4))          # 140 voices

    def make():
        return pg.MixPE(*[c5_voice(pg, i) for i in idx])

    blocks = [(0, 6000), (6000, 2048), (8048, 4097), (30000, 1000)]
    fused = make()
    got = _render_blocks(fused, 48000, blocks)
    assert fused._bank and fused._bank.k == len(idx)
    monkeypatch.setattr(voice_bank, "FUSED_VOICE_MIN", 10 ** 9)
    monkeypatch.setattr(voice_bank, "WIDE_SUPERSAW", False)           # the two-launch bank: k_blitsaw's own samples
    plain = make()
    want = _render_blocks(plain, 48000, blocks)
    for a, b in zip(got, want):
        assert a.shape == b.shape
        peak = float(np.max(np.abs(b)))
        assert float(np.max(np.abs(a.astype(np.float64) - b))) <= (1e-6 if wide else 2e-7) * peak
        if not wide:
            assert np.mean(a != b) < 1e-3


@pytest.mark.parametrize("gain_in_chain", [False, True])
def test_envelopes_one_block_ahead_change_nothing(monkeypatch, gain_in_chain):
    """A C5 bank streamed in equal blocks walks block k+1's envelopes on the side stream while block k is mixed
    (voice_bank.ENVELOPE_AHEAD); a seek or a different block length puts the envelope states back.  Same kernels, same
    order of operations per voice: bit for bit the bank without it."""
    from pygmu2_amd import voice_bank
    pg.set_sample_rate(48000)
    blocks = ([(i * 6000, 6000) for i in range(5)] + [(30_000, 4096), (34_096, 4096), (38_192, 4096)]    # a new block length
              + [(100_000, 6000), (106_000, 6000), (112_000, 6000)]                                         # a seek
              + [(0, 6000), (6000, 6000)])                                                                   # and back to the start

    monkeypatch.setattr(voice_bank, "FUSE_GAIN_IN_CHAIN", gain_in_chain)    # (x gain inside pgx_blitsaw_biquad_wide: same float32 products)
    monkeypatch.setattr(voice_bank, "VOICE_TILES", False)      # the layered path: its kernels do not depend on what runs ahead (the
                                                               # on-chip mix, a tolerance path in windows: tests/test_gpu_voice_tiles.py)

    def run(ahead):
        monkeypatch.setattr(voice_bank, "ENVELOPE_AHEAD", ahead)
        mix = pg.MixPE(*[c5_voice(pg, i) for i in range(0, 512, 3)])
        r = pg.NullRenderer(sample_rate=48000)
        r.set_source(mix)
        r.start()
        outs, armed = [], 0
        for s, n in blocks:
            outs.append(mix.render(s, n).data.copy())
            armed += mix._voice_bank().root.children["gain"].ahead is not None
        r.stop()
        return outs, armed

    got, armed = run(True)
    want, none = run(False)
    assert armed >= 7 and none == 0
    for a, b in zip(got, want):
        assert np.array_equal(a, b), float(np.max(np.abs(a - b)))


def test_ladder_bank_windows_hand_out_the_block_by_block_samples(monkeypatch):
    """A C4 bank streamed in equal blocks renders 2, 4, 8 blocks per ladder launch and hands out rows of that window
    (voice_bank.LADDER_WINDOWS); a seek inside a window, another block length and a restart put the states back.
    Against the same bank block by block: the time-segmented ladder's bound (<= 1e-6 of peak; its segments fall
    elsewhere in a longer render)."""
    from pygmu2_amd import voice_bank
    from pygmu2_amd.sharding import c4_voice
    pg.set_sample_rate(48000)
    n = 8192
    blocks = ([(i * n, n) for i in range(9)]                      # windows of 2, 4, (8: left after 2 of its blocks)
              + [(9 * n + 100, n), (10 * n + 100, n), (11 * n + 100, n)]          # a seek inside the window
              + [(12 * n + 100, 5000), (12 * n + 5100, 5000), (12 * n + 10100, 5000)]    # another block length
              + [(0, n), (n, n), (2 * n, n)])                                    # and from the start again

    def run(windows):
        monkeypatch.setattr(voice_bank, "LADDER_WINDOWS", windows)
        mix = pg.MixPE(*[c4_voice(pg, i) for i in range(0, 64, 4)])
        r = pg.NullRenderer(sample_rate=48000)
        r.set_source(mix)
        r.start()
        outs, opened = [], 0
        for s, m in blocks:
            outs.append(mix.render(s, m).data.copy())
            opened += mix._voice_bank().root.win is not None
        r.stop()
        return outs, opened

    got, opened = run(True)
    want, none = run(False)
    assert opened >= 8 and none == 0
    for a, b in zip(got, want):
        peak = float(np.max(np.abs(b)))
        assert a.shape == b.shape and float(np.max(np.abs(a.astype(np.float64) - b))) <= 1e-6 * peak


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_oscillator_filter_chain_random_banks(monkeypatch, seed):
    """pgx_blitsaw_biquad_wide on random banks -- 128 .. 300 voices, oscillators from 20 Hz to 9 kHz, every RBJ mode,
    corner frequencies from 40 Hz to 18 kHz, q from 0.3 to 12, block lengths from 1 frame to 48 000 with odd tails and
    a seek -- against the two-launch bank (k_blitsaw's samples through pgx_biquad_const): <= 2e-6 of the peak."""
    from pygmu2_amd import voice_bank
    from pygmu2_amd.biquad_pe import BiquadMode
    rng = np.random.default_rng(4200 + seed)
    pg.set_sample_rate(48000)
    count = [128, 131, 200, 300][seed - 1]
    modes = list(BiquadMode)
    spec = [(float(np.exp(rng.uniform(np.log(20.0), np.log(9000.0)))), modes[int(rng.integers(0, len(modes)))],
             float(np.exp(rng.uniform(np.log(40.0), np.log(18000.0)))), float(np.exp(rng.uniform(np.log(0.3), np.log(12.0)))),
             float(rng.uniform(-12.0, 12.0))) for _ in range(count)]
    mode0 = spec[0][1]                                     # one bank: same structure, the mode is part of it

    def make():
        return pg.MixPE(*[pg.BiquadPE(pg.BlitSawPE(f), frequency=fc, q=q, mode=mode0, gain_db=g)
                          for f, _, fc, q, g in spec])

    sizes = [int(rng.choice([1, 15, 17, 4095, 4097, 12_289, 48_000])) for _ in range(5)]
    blocks, pos = [], 0
    for i, n in enumerate(sizes):
        if i == 3:
            pos += 123_457
        blocks.append((pos, n))
        pos += n
    monkeypatch.setattr(voice_bank, "WIDE_SUPERSAW", True)
    fused = make()
    got = _render_blocks(fused, 48000, blocks)
    assert fused._bank and fused._bank.k == count
    used = fused._bank.root.children["source"].wide()
    monkeypatch.setattr(voice_bank, "FUSED_VOICE_MIN", 10 ** 9)
    monkeypatch.setattr(voice_bank, "WIDE_SUPERSAW", False)
    want = _render_blocks(make(), 48000, blocks)
    if not used:
        pytest.skip("an oscillator the rotation form excludes: the bank kept the eight-frames-per-thread kernel")
    peak = max(float(np.max(np.abs(b))) for b in want)
    for (s, n), a, b in zip(blocks, got, want):
        assert a.shape == b.shape
        err = float(np.max(np.abs(a.astype(np.float64) - b)))
        assert err <= 2e-6 * peak, (s, n, err, peak, mode0)


@pytest.mark.parametrize("count", [4, 17, 64, 200, 256])
def test_small_bank_chain_in_time_segments(monkeypatch, count):
    """pgx_blitsaw_biquad_wide_seg (round 4: a rank's share of C5): oscillator -> filter of a small bank as ONE launch in
    concurrent time segments -- the oscillator's integrator level on entering a segment from its closed form, the
    filter from a warm-up of settle_frames -- against the same bank without it (segmented oscillator bank + batched
    settled biquad) and against the two-launch bank on k_blitsaw's own samples; states carried over blocks (read from
    one buffer, written to the other), odd tails, a seek."""
    from pygmu2_amd import voice_bank
    from pygmu2_amd.biquad_pe import BiquadMode
    rng = np.random.default_rng(9100 + count)
    pg.set_sample_rate(48000)
    mode = list(BiquadMode)[count % 3]
    spec = [(float(np.exp(rng.uniform(np.log(25.0), np.log(6000.0)))),
             float(np.exp(rng.uniform(np.log(300.0), np.log(12000.0)))), float(rng.uniform(0.5, 4.0)))
            for _ in range(count)]

    def make():
        return pg.MixPE(*[pg.BiquadPE(pg.BlitSawPE(f), frequency=fc, q=q, mode=mode) for f, fc, q in spec])

    blocks = [(0, 48_000), (48_000, 48_000), (96_000, 20_001), (116_001, 12_289), (400_000, 48_000), (448_000, 48_000)]
    monkeypatch.setattr(voice_bank, "BANK_WINDOWS", False)
    seg = make()
    got = _render_blocks(seg, 48000, blocks)
    node = seg._bank.root
    assert isinstance(node, voice_bank._BiquadNode)
    if count < 256:                                  # (256 voices: one workgroup per CU already -- a single segment)
        assert node._chain_segments(48_000) > 1, "the segmented chain was expected"
    monkeypatch.setattr(voice_bank, "SEGMENTED_CHAIN", False)
    same = _render_blocks(make(), 48000, blocks)
    monkeypatch.setattr(voice_bank, "FUSED_VOICE_MIN", 10 ** 9)
    monkeypatch.setattr(voice_bank, "WIDE_SUPERSAW", False)
    want = _render_blocks(make(), 48000, blocks)
    peak = max(float(np.max(np.abs(b))) for b in want)
    for (s, n), a, b, c in zip(blocks, got, same, want):
        assert a.shape == c.shape
        assert float(np.max(np.abs(a.astype(np.float64) - b))) <= 1e-6 * peak, (s, n, "against the two-kernel wide path")
        assert float(np.max(np.abs(a.astype(np.float64) - c))) <= 2e-6 * peak, (s, n, "against the two-launch bank")
