"""GPU: two and three REAL ranks (processes) of a sharded mix on one card, exchanging through gloo on host payloads --
the rank logic of ShardedMixPE with the device banks underneath: every rank pulls the same random sequence (streams of
equal blocks, seeks, steps back, odd lengths), the SuperSaw and SuperSaw -> ladder shares render windows of 2, 4, 8
blocks and reduce each window in ONE collective, C5 shares reduce block by block -- and every rank ends up with the full
mix.  A rank that issued a different sequence of collectives (another size, one more, one less) would fail or hang gloo:
the run itself is the check that all ranks decide alike."""

import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PGX_ROOT"])
rank, world, kind, total, seed, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
import torch.distributed as dist
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
import pygmu2_amd as pg
from pygmu2_amd.sharding import ShardedMixPE, TorchReducer, mix_voice_factory
pg.set_sample_rate(48000)
make = mix_voice_factory(kind)[0]
root = ShardedMixPE([make(pg, i) for i in range(total)], rank, world)
sizes = []
pulls = [tuple(int(v) for v in p) for p in np.load(os.path.join(out_dir, "pulls.npy"))]
r = pg.NullRenderer(48000); r.set_source(root); r.start()
got = []
for s, n in pulls:
    got.append(root.render(s, n).data.copy())
    if isinstance(root._reducer, TorchReducer) and not hasattr(root._reducer, "_counted"):
        inner = root._reducer.all_reduce
        def counted(snippet, inner=inner):
            sizes.append(snippet.duration)
            return inner(snippet)
        root._reducer.all_reduce = counted
        root._reducer._counted = True
r.stop()
assert isinstance(root._reducer, TorchReducer) and not root._reducer.on_device
np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.concatenate(got))
np.save(os.path.join(out_dir, f"sizes{rank}.npy"), np.array(sizes))
dist.barrier()
dist.destroy_process_group()
print("RANK_OK")
'''


def _pulls(rng):
    n = int(rng.choice([4096, 12_288]))
    pos, pulls = 0, []
    for _ in range(int(rng.integers(18, 26))):
        what = rng.random()
        if what < 0.07:
            pos += int(rng.integers(1, 50_000))
        elif what < 0.11:
            pos = max(0, pos - int(rng.integers(1, 3 * n)))
        size = n if rng.random() < 0.92 else int(rng.choice([17, 5000, 2 * n]))
        pulls.append((pos, size))
        pos += size
    return pulls


@pytest.mark.parametrize("world,kind,total,seed", [(2, "supersaw", 48, 1), (3, "supersaw", 40, 2), (2, "c4", 12, 3),
                                                   (3, "c5", 30, 4)])
def test_every_rank_gets_the_full_mix(tmp_path, world, kind, total, seed):
    import pygmu2_amd as pg
    from pygmu2_amd import voice_bank
    from pygmu2_amd.sharding import mix_voice_factory
    pulls = _pulls(np.random.default_rng(seed))
    np.save(tmp_path / "pulls.npy", np.array(pulls, dtype=np.int64))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PGX_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), kind, str(total), str(seed), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=600)
            outs.append(out.decode()[-3000:])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, out in zip(procs, outs):
        assert p.returncode == 0 and "RANK_OK" in out, out
    # the full mix, unsharded, every pull rendered when it is asked for
    pg.set_sample_rate(48000)
    keep = {k: getattr(voice_bank, k) for k in ("BANK_WINDOWS", "LADDER_WINDOWS", "ENVELOPE_AHEAD", "PREFETCH_SUPERSAW_VOICES",
                                                "PREFETCH_LADDER_INPUT")}
    for k in keep:
        setattr(voice_bank, k, False)
    try:
        make = mix_voice_factory(kind)[0]
        full = pg.MixPE(*[make(pg, i) for i in range(total)])
        r = pg.NullRenderer(48000)
        r.set_source(full)
        r.start()
        want = np.concatenate([full.render(s, n).data.copy() for s, n in pulls])
        r.stop()
    finally:
        for k, v in keep.items():
            setattr(voice_bank, k, v)
    peak = float(np.max(np.abs(want)))
    sizes0 = np.load(tmp_path / "sizes0.npy")
    for r_ in range(world):
        got = np.load(tmp_path / f"rank{r_}.npy")
        assert got.shape == want.shape
        assert float(np.max(np.abs(got.astype(np.float64) - want))) <= 1e-5 * peak, r_
        assert np.array_equal(np.load(tmp_path / f"sizes{r_}.npy"), sizes0)          # the same collectives, in the same order
    if kind in ("supersaw", "c4"):
        n = max(set(m for _, m in pulls), key=[m for _, m in pulls].count)
        assert int(sizes0.max()) >= 4 * n, sizes0                                    # windows were reduced whole
    else:
        assert len(sizes0) >= len(pulls) - 1 and int(sizes0.max()) <= max(m for _, m in pulls)
