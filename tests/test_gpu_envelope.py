"""
GPU: EnvelopePE's attack/release follower (pgx_envelope, time-parallel Newton rounds over windows of
1024..8192 samples) against the oracle's literal loop (orc_envelope_ar, envelope_pe.py:259-271) on inputs
that exercise every path: long regimes (periodic input), regime flips every few samples (noise: most rounds),
instant attack / instant release (zero slopes), silence, block lengths around every window size, stereo,
state carried across blocks, RMS detection (separate detector pass).
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


def _signals(n):
    rng = np.random.default_rng(12)
    t = np.arange(n) / 44100.0
    burst = np.sin(2 * np.pi * 220.0 * t) * np.exp(-((t % 0.5) * 6.0))
    return {
        "sine": 0.8 * np.sin(2 * np.pi * 220.0 * t),
        "noise": rng.standard_normal(n) * 0.3,
        "bursts": burst,
        "silence_then_step": np.concatenate([np.zeros(n // 3), np.full(n - n // 3, 0.5)]),
        "slow_am": np.sin(2 * np.pi * 3000.0 * t) * (0.5 + 0.5 * np.sin(2 * np.pi * 2.0 * t)),
    }


@pytest.mark.parametrize("name", ["sine", "noise", "bursts", "silence_then_step", "slow_am"])
@pytest.mark.parametrize("attack,release", [(0.005, 0.05), (0.0, 0.03), (0.05, 0.001), (0.02, 0.0)])
def test_attack_release_follower_matches_oracle(name, attack, release):
    from oracle import pe_oracle as O
    import pygmu2_amd as pg
    pg.set_sample_rate(44100)
    n = 50_000
    mono = _signals(n)[name]
    x = np.stack([mono, np.roll(mono, 777) * 0.5], axis=1).astype(np.float32)
    pe = pg.EnvelopePE(pg.ArrayPE(x), attack=attack, release=release)
    r = pg.NullRenderer(sample_rate=44100)
    r.set_source(pe)
    r.start()
    sizes = [1024, 63, 64, 65, 1, 1025, 2048, 2049, 4096, 4097, 8192, 5, 27271]      # sums to n
    pos, got = 0, []
    for s in sizes:
        got.append(pe.render(pos, s).data)
        pos += s
    r.stop()
    got = np.concatenate(got)
    st = O.envelope_state()
    want = O.envelope(st, x, attack=attack, release=release, mode="peak", sr=44100)
    peak = float(np.max(np.abs(want))) or 1.0
    err = float(np.max(np.abs(got.astype(np.float64) - want)))
    assert err <= REL_TOL * peak + 1e-7, (name, attack, release, err, peak)
    assert np.all(got >= 0.0)


@pytest.mark.parametrize("mode", ["peak", "rms"])
def test_follower_long_block_and_detectors(mode):
    """One 300 000-frame render (74 windows of the widest kernel) and the RMS detector in front of the follower."""
    from oracle import pe_oracle as O
    import pygmu2_amd as pg
    pg.set_sample_rate(48000)
    n = 300_000
    rng = np.random.default_rng(5)
    t = np.arange(n) / 48000.0
    x = (np.sin(2 * np.pi * 97.0 * t) * (0.2 + 0.8 * (np.sin(2 * np.pi * 0.7 * t) > 0)) +
         0.05 * rng.standard_normal(n)).astype(np.float32).reshape(-1, 1)
    pe = pg.EnvelopePE(pg.ArrayPE(x), attack=0.002, release=0.12, mode=pg.DetectionMode(mode))
    r = pg.NullRenderer(sample_rate=48000)
    r.set_source(pe)
    r.start()
    got = pe.render(0, n).data
    r.stop()
    want = O.envelope(O.envelope_state(), x, attack=0.002, release=0.12, mode=mode, sr=48000)
    peak = float(np.max(np.abs(want)))
    err = float(np.max(np.abs(got.astype(np.float64) - want)))
    assert err <= REL_TOL * peak + 1e-7, (mode, err, peak)


@pytest.mark.parametrize("attack,release", [(0.005, 0.05), (0.001, 2.0), (0.3, 0.0005), (0.0, 0.03)])
@pytest.mark.parametrize("name", ["sine", "noise", "bursts", "silence_then_step"])
def test_all_windows_at_once_matches_oracle_and_carries_state(name, attack, release):
    """Blocks of >= 131 072 frames take the form that solves every 8192-sample window concurrently (Newton rounds
    over the windows' entry levels, one launch per round): against the oracle's literal loop, over two such
    blocks and a short one in a row (carried level), with a slow release (a window barely forgets its entry)."""
    from oracle import pe_oracle as O
    import pygmu2_amd as pg
    pg.set_sample_rate(44100)
    n = 700_000
    mono = _signals(n)[name]
    x = np.stack([mono, np.roll(mono, 1234) * 0.25], axis=1).astype(np.float32)
    pe = pg.EnvelopePE(pg.ArrayPE(x), attack=attack, release=release)
    r = pg.NullRenderer(sample_rate=44100)
    r.set_source(pe)
    r.start()
    got = np.concatenate([pe.render(s, m).data for s, m in ((0, 300_001), (300_001, 4000), (304_001, 395_999))])
    r.stop()
    st = O.envelope_state()
    want = O.envelope(st, x, attack=attack, release=release, mode="peak", sr=44100)
    peak = float(np.max(np.abs(want))) or 1.0
    assert float(np.max(np.abs(got.astype(np.float64) - want))) <= REL_TOL * peak + 1e-7


def test_all_windows_form_falls_back_when_its_rounds_run_out(tmp_path):
    """One round is never enough (the first round renders every window from the same guess): the block must come
    out of the sequential kernel, identical to a render that never tried."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PGX_ROOT"])
import pygmu2_amd as pg
pg.set_sample_rate(44100)
t = np.arange(400_000) / 44100.0
x = (0.7 * np.sin(2 * np.pi * 110.0 * t) * (0.5 + 0.5 * np.sin(2 * np.pi * 1.5 * t))).astype(np.float32)
pe = pg.EnvelopePE(pg.ArrayPE(x), attack=0.004, release=0.08)
r = pg.NullRenderer(44100); r.set_source(pe); r.start()
a = np.concatenate([pe.render(0, 250_000).data, pe.render(250_000, 150_000).data])
r.stop()
np.save(os.environ["PGX_OUT"], a)
""")
    outs = []
    for rounds in ("1", "8"):
        out = str(tmp_path / f"r{rounds}.npy")
        env = dict(os.environ, PGX_ROOT=root, PGX_OUT=out, PGX_ENV_MW_ROUNDS=rounds)
        p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(np.load(out))
    assert outs[0].shape == outs[1].shape
    assert float(np.max(np.abs(outs[0] - outs[1]))) <= 1e-6 * float(np.max(np.abs(outs[1])))
