"""GPU: two and three REAL ranks (processes) of a sharded mix on one card, exchanging through gloo on host payloads --
the rank logic of ShardedMixPE with the device banks underneath: every rank pulls the same random sequence (streams of
equal blocks, seeks, steps back, odd lengths), the SuperSaw and SuperSaw -> ladder shares render windows of 2, 4, 8
blocks and reduce each window in ONE collective, as do C5 shares of 16 voices and more (voices mixed on chip); smaller C5
shares reduce block by block -- and every rank ends up with the full
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
                                                   (3, "c5", 30, 4), (3, "c5", 60, 5), (2, "c5", 70, 6)])
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
    if kind in ("supersaw", "c4") or (kind == "c5" and total // world >= voice_bank.VOICE_TILES_MIN_VOICES):
        # (C5 shares of 16 voices and more mix their voices on chip, in windows: one collective each, like the others)
        n = max(set(m for _, m in pulls), key=[m for _, m in pulls].count)
        assert int(sizes0.max()) >= 4 * n, sizes0                                    # windows were reduced whole
    else:
        assert len(sizes0) >= len(pulls) - 1 and int(sizes0.max()) <= max(m for _, m in pulls)


BENCH_WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["PGX_ROOT"])
rank, world, kind, voices, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
import torch.distributed as dist
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
import torch
import pygmu2_amd as pg
from pygmu2_amd.sharding import bench_voice_mix

class Dist:                      # what bench.py's Dist offers a workload, carried by the gloo group
    enabled = True
    def __init__(self): self.world, self.rank = world, rank
    def barrier(self): dist.barrier()
    def max_over_ranks(self, v):
        t = torch.tensor([float(v)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

dt, frames, name, info = bench_voice_mix(pg, Dist(), 6, 3, voices=voices, block=12_288, config=kind)
with open(os.path.join(out_dir, f"info{rank}.json"), "w") as f:
    json.dump({"dt": dt, "frames": frames, "name": name, "info": info}, f)
dist.barrier()
dist.destroy_process_group()
print("RANK_OK")
'''


@pytest.mark.parametrize("kind,voices", [("supersaw", 24), ("c5", 16), ("c4", 8)])
def test_the_bench_workload_with_two_real_ranks(tmp_path, kind, voices):
    """bench.py's sharded workload (sharding.bench_voice_mix) with two real rank processes on one card over gloo: the
    N > 1 parts of it that a one-GPU bench run never executes -- the sharded-against-unsharded parity figure measured
    inside the run (rank 0 renders the full mix as well), the collective counts, the per-rank render time."""
    import json
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "w.py"
    script.write_text(BENCH_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PGX_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), kind, str(voices), str(tmp_path)],
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
    infos = [json.load(open(tmp_path / f"info{r}.json")) for r in range(world)]
    i0 = infos[0]["info"]
    assert i0["owned"] == voices // 2 and infos[1]["info"]["owned"] == voices - voices // 2
    assert i0["sharded_vs_unsharded_max_err_over_peak"] is not None and i0["sharded_vs_unsharded_max_err_over_peak"] <= 1e-6
    assert infos[1]["info"]["sharded_vs_unsharded_max_err_over_peak"] is None       # rank 0 alone renders the full mix
    assert i0["collectives_in_timed_region"] == infos[1]["info"]["collectives_in_timed_region"] >= 1
    assert i0["agreement_checks"] >= 1 and i0["render_ms"] > 0 and infos[0]["dt"] == infos[1]["dt"] > 0
