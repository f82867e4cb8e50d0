"""GPU: BASELINE.json's sharded configurations at their FULL voice counts and block size, through the property the
sharding rests on: the mix is a plain sum (mix_pe.py:91-94), so the shares rendered by the G ranks (inputs
i = r mod G, each with its own state) add up to the unsharded mix.  One GPU plays every rank in turn; this also
runs the kernel variants a rank's smaller share selects (several workgroups per oscillator below 128 voices, a
workgroup per envelope in the ADSR walk) against the bank-wide ones at full size.  Two consecutive 48 000-frame
blocks, so carried state is part of it."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BLOCK = 48_000


def _render(pg, voices, blocks=2):
    root = pg.MixPE(*voices) if len(voices) > 1 else voices[0]
    r = pg.NullRenderer(sample_rate=48000)
    r.set_source(root)
    r.start()
    out = [root.render(i * BLOCK, BLOCK).data.astype(np.float64) for i in range(blocks)]
    r.stop()
    return np.concatenate(out)


@pytest.mark.parametrize("config,n_voices,world", [("c5", 512, 8), ("c4", 64, 4), ("supersaw", 512, 8)])
def test_rank_shares_add_up_to_the_full_mix(config, n_voices, world):
    import pygmu2_amd as pg
    from pygmu2_amd.sharding import c4_voice, c5_voice, shard_indices, supersaw_voice
    pg.set_sample_rate(48000)
    make = {"c5": c5_voice, "c4": c4_voice, "supersaw": supersaw_voice}[config]
    full = _render(pg, [make(pg, i) for i in range(n_voices)])
    total = np.zeros_like(full)
    for rank in range(world):
        total += _render(pg, [make(pg, i) for i in shard_indices(n_voices, rank, world)])
    peak = float(np.max(np.abs(full)))
    assert peak > 0.1 and np.all(np.isfinite(full))
    # float32 sums in a different grouping: n_voices terms of magnitude <= 1
    assert float(np.max(np.abs(total - full))) <= 1e-5 * peak, (config, float(np.max(np.abs(total - full))), peak)
