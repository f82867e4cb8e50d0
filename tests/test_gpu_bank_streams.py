"""
GPU: voice banks pulled the way a renderer pulls them -- long streams of equal blocks with seeks, odd lengths, steps back
and restarts thrown in -- with everything that runs ahead of the caller switched ON (bank windows of 2, 4, 8 blocks,
ladder windows, envelopes and oscillators one block ahead, mix-level windows as ShardedMixPE asks for them) against the
same bank with all of it switched OFF (every pull rendered when it is asked for).  The features only reorder work: what
the caller gets may differ where a longer render cuts its time segments elsewhere (<= 1e-6 of the peak), nowhere else.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SWITCHES = ("BANK_WINDOWS", "LADDER_WINDOWS", "ENVELOPE_AHEAD", "PREFETCH_SUPERSAW_VOICES", "PREFETCH_LADDER_INPUT")


def _voices(pg, kind, count, rng):
    from pygmu2_amd.sharding import c4_voice, c5_voice, supersaw_voice
    if kind == "supersaw":
        nv = int(rng.choice([1, 3, 7]))
        return [pg.SuperSawPE(55.0 * 2 ** (i / 24.0), voices=nv, detune_cents=20.0, seed=i) for i in range(count)]
    if kind == "c4":
        return [c4_voice(pg, i) for i in range(count)]
    if kind == "c5":
        return [c5_voice(pg, 3 * i) for i in range(count)]
    if kind == "comb":
        return [pg.CombPE(pg.BlitSawPE(55.0 * 2 ** (i / 12.0)), frequency=110.0 * 2 ** (i / 24.0), feedback=0.7)
                for i in range(count)]
    raise KeyError(kind)


def _pulls(rng):
    n = int(rng.choice([4096, 12_288, 48_000]))
    pos, pulls = 0, []
    for _ in range(int(rng.integers(14, 30))):
        what = rng.random()
        if what < 0.06:
            pos += int(rng.integers(1, 100_000))                 # a seek forward
        elif what < 0.10:
            pos = max(0, pos - int(rng.integers(1, 3 * n)))      # a step back
        elif what < 0.13:
            pos = 0                                              # from the top
        size = n if rng.random() < 0.93 else int(rng.choice([1, 17, 5000, 2 * n]))
        pulls.append((pos, size))
        pos += size
    return pulls


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PGX_FUZZ_BANK_STREAMS", "16"))))
def test_running_ahead_changes_nothing(monkeypatch, seed):
    import pygmu2_amd as pg
    from pygmu2_amd import voice_bank
    rng = np.random.default_rng(64_000 + seed)
    kind = str(rng.choice(["supersaw", "supersaw", "c4", "c5", "comb"]))
    count = int(rng.choice({"supersaw": [8, 40, 64, 130, 260], "c4": [4, 8, 20], "c5": [6, 64, 130], "comb": [8, 40]}[kind]))
    mix_windows = kind in ("supersaw", "c4") and rng.random() < 0.4        # what a rank of a sharded mix asks for
    pulls = _pulls(rng)
    pg.set_sample_rate(48000)

    def run(ahead):
        for name in SWITCHES:
            monkeypatch.setattr(voice_bank, name, ahead)
        mix = pg.MixPE(*_voices(pg, kind, count, np.random.default_rng(seed)))
        if ahead and mix_windows:
            mix.__dict__["_mix_windows"] = True
        r = pg.NullRenderer(sample_rate=48000)
        r.set_source(mix)
        r.start()
        assert mix._voice_bank()
        outs = [mix.render(s, n).data.copy() for s, n in pulls]
        r.stop()
        return outs

    got, want = run(True), run(False)
    peak = max(float(np.max(np.abs(w))) for w in want) or 1.0
    for i, ((s, n), a, b) in enumerate(zip(pulls, got, want)):
        assert a.shape == b.shape
        err = float(np.max(np.abs(a.astype(np.float64) - b)))
        assert err <= 1e-6 * peak, (kind, count, mix_windows, i, pulls[max(0, i - 2):i + 1], err, peak)
