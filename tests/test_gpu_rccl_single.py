"""GPU: the RCCL leg of ShardedMixPE on 1-rank communicators -- the library's own entry points
(pgx_comm_init / pgx_allreduce_sum / pgx_allreduce_wait: the product path, no torch in the process), the
deferred wait and buffer lifetimes of the pipelined use, the alternative torch.distributed binding, and
bench.py's launch glue under torch.distributed.run."""

import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PGX_ROOT"])
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
import pygmu2_amd as pg
from pygmu2_amd.sharding import ShardedMixPE, TorchReducer, c5_voice
pg.set_sample_rate(48000)
voices = [c5_voice(pg, i) for i in range(6)]
red = TorchReducer()
assert red.on_device
root = ShardedMixPE(voices, 0, 1)
root._world = 2                      # force the reduction path; with one rank the sum is the identity
root._reducer = red
r = pg.NullRenderer(48000); r.set_source(root); r.start()
a = [root.render(i * 4096, 4096).data for i in range(3)]
r.stop()
plain = pg.MixPE(*[c5_voice(pg, i) for i in range(6)])
r = pg.NullRenderer(48000); r.set_source(plain); r.start()
b = [plain.render(i * 4096, 4096).data for i in range(3)]
r.stop()
for x, y in zip(a, b):
    assert np.array_equal(x, y), float(np.max(np.abs(x - y)))
# pipelined use: blocks are rendered back to back and dropped unread while their all-reduce may still be in
# flight (the Snippet's `ready` hook must keep the buffer out of the pool until the collective is ordered)
root2 = ShardedMixPE([c5_voice(pg, i) for i in range(6)], 0, 1)
root2._world = 2
root2._reducer = red
r = pg.NullRenderer(48000); r.set_source(root2); r.start()
keep = None
for i in range(3):
    keep = root2.render(i * 4096, 4096)
assert keep._ready is not None       # nothing has forced the last reduce yet
assert np.array_equal(keep.data, b[2])
assert keep._ready is None
r.stop()
# the tensor view really aliases the library buffer
from pygmu2_amd import device
buf = device.DeviceBuffer.from_host(np.arange(8, dtype=np.float32))
t = torch.as_tensor(buf, device="cuda")
t += 1
torch.cuda.synchronize()
assert buf.to_host().tolist() == [1, 2, 3, 4, 5, 6, 7, 8]
dist.destroy_process_group()
print("RCCL_SINGLE_OK")
'''


ABI_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PGX_ROOT"])
import pygmu2_amd as pg
from pygmu2_amd import comm, device
from pygmu2_amd.sharding import ShardedMixPE, RcclReducer, default_reducer, c5_voice
device.ensure_init()
assert not comm.initialised()
comm.init(0, 1, comm.unique_id())
assert comm.info() == (0, 1)
assert comm.reduce_scalar(3.5, "sum") == 3.5 and comm.reduce_scalar(-2.0, "max") == -2.0
pg.set_sample_rate(48000)
root = ShardedMixPE([c5_voice(pg, i) for i in range(6)], 0, 1)
root._world = 2                      # force the reduction path; with one rank the sum is the identity
r = pg.NullRenderer(48000); r.set_source(root); r.start()
a = [root.render(i * 4096, 4096).data for i in range(3)]
r.stop()
assert isinstance(root._reducer, RcclReducer)
st = comm.stats()
assert st["issued"] == 3 and st["checks"] == 3 and st["hash"] != 0, st     # the first collectives are each checked
plain = pg.MixPE(*[c5_voice(pg, i) for i in range(6)])
r = pg.NullRenderer(48000); r.set_source(plain); r.start()
b = [plain.render(i * 4096, 4096).data for i in range(3)]
r.stop()
for x, y in zip(a, b):
    assert np.array_equal(x, y), float(np.max(np.abs(x - y)))
# pipelined: blocks dropped unread while their all-reduce may still be in flight; 200 blocks also walk the
# ticket ring (64 event slots) several times
root2 = ShardedMixPE([c5_voice(pg, i) for i in range(6)], 0, 1)
root2._world = 2
r = pg.NullRenderer(48000); r.set_source(root2); r.start()
keep = None
for i in range(200):
    keep = root2.render(i * 512, 512)
assert keep._ready is not None       # nothing has forced the last reduce yet
tail = keep.data
assert keep._ready is None
r.stop()
st = comm.stats()
assert st["issued"] == 203 and st["checks"] == 8 + (203 // 16 - 0), st      # first 8, then every 16th ticket
assert device.load_library().pgx_comm_quiesce(5000) == 0
r = pg.NullRenderer(48000); r.set_source(plain); r.start()
want = np.concatenate([plain.render(i * 512, 512).data for i in range(200)])[-512:]
r.stop()
assert np.max(np.abs(tail - want)) <= 1e-6 * max(1e-3, float(np.max(np.abs(want))))
# an idle rank (owns nothing) and a one-voice rank issue the same collective; the local payload is left alone
cached = pg.CachePE(pg.SinePE(frequency=300.0))
solo = ShardedMixPE([cached, pg.SinePE(frequency=500.0)], 0, 2, reducer=RcclReducer())
first = solo.render(0, 256).data.copy()
again = solo.render(0, 256).data
assert np.array_equal(first, again) and np.array_equal(first, pg.SinePE(frequency=300.0).render(0, 256).data)
idle = ShardedMixPE([pg.SinePE(frequency=300.0), pg.SinePE(frequency=500.0)], 2, 3, reducer=RcclReducer())
assert not np.any(idle.render(0, 128).data)
# a rank's share of the sharded SuperSaw mix at 8 ranks: 64 instances, oscillators one block ahead on the main
# stream, voice sum + mix on the side stream, the all-reduce of every block behind the next block's oscillators;
# a seek in the middle takes the block rendered ahead back
from pygmu2_amd.sharding import supersaw_voice, shard_indices
mine = list(shard_indices(512, 0, 8))
share = ShardedMixPE([supersaw_voice(pg, i) for i in range(512)], 0, 8, reducer=RcclReducer())
assert len(share.owned) == 64
pulls = [(0, 48000), (48000, 48000), (96000, 48000), (500000, 48000), (548000, 4096), (552096, 48000)]
r = pg.NullRenderer(48000); r.set_source(share); r.start()
got = [share.render(s, n).data.copy() for s, n in pulls]
r.stop()
from pygmu2_amd import look_ahead
plain64 = pg.MixPE(*[supersaw_voice(pg, i) for i in mine])
plain64._bank = False                # per-voice rendering, no bank, no overlap ...
look_ahead.set_enabled(False)        # ... and block by block: a window cuts the phase sums differently (1 ulp here and there)
r = pg.NullRenderer(48000); r.set_source(plain64); r.start()
want64 = [plain64.render(s, n).data.copy() for s, n in pulls]
r.stop()
look_ahead.set_enabled(True)
for x, y in zip(got, want64):
    # the share is rendered by the time-segmented fused bank (pgx_supersaw_wide: integrator carries from the closed
    # form, phases as products, voices summed in float64): the per-voice samples to one or two float32 ulps
    assert float(np.max(np.abs(x.astype(np.float64) - y))) <= 1e-6 * float(np.max(np.abs(y)))
# a stream of equal blocks: the bank renders windows of 2, 4, 8 blocks and each window is ONE collective (16 blocks:
# block 0 alone, then windows of 2, 4, 8 and 8 -- five collectives); rows dropped unread do not wait for it
share = ShardedMixPE([supersaw_voice(pg, i) for i in range(512)], 0, 8, reducer=RcclReducer())
sizes = []
inner = share._reducer.all_reduce
def counted(snippet):
    sizes.append(snippet.duration)
    return inner(snippet)
share._reducer.all_reduce = counted
r = pg.NullRenderer(48000); r.set_source(share); r.start()
rows, keep = [], None
for i in range(16):
    keep = share.render(i * 48000, 48000)
    if i in (0, 1, 2, 6, 9, 15):
        rows.append((i, keep.data.copy()))
    elif i == 12:
        assert keep._ready is not None
r.stop()
assert sizes == [48000, 96000, 192000, 384000, 384000], sizes
look_ahead.set_enabled(False)
r = pg.NullRenderer(48000); r.set_source(plain64); r.start()
for i in range(16):
    y = plain64.render(i * 48000, 48000).data
    for j, x in rows:
        if j == i:
            assert float(np.max(np.abs(x.astype(np.float64) - y))) <= 1e-6 * float(np.max(np.abs(y))), i
r.stop()
look_ahead.set_enabled(True)
# the same for a rank's share of C4 (8 of the 64 SuperSaw -> ladder instances): ShardedMixPE asks the bank for windows at
# the level of the mix (the ladder node's own windows hand out rows BELOW the mix: a collective per block)
from pygmu2_amd.sharding import c4_voice
share4 = ShardedMixPE([c4_voice(pg, i) for i in range(64)], 0, 8, reducer=RcclReducer())
assert len(share4.owned) == 8
sizes4 = []
inner4 = share4._reducer.all_reduce
def counted4(snippet):
    sizes4.append(snippet.duration)
    return inner4(snippet)
share4._reducer.all_reduce = counted4
r = pg.NullRenderer(48000); r.set_source(share4); r.start()
got4 = [share4.render(i * 48000, 48000).data.copy() for i in range(8)]
r.stop()
assert sizes4 == [48000, 96000, 192000, 384000], sizes4
plain8 = pg.MixPE(*[c4_voice(pg, i) for i in shard_indices(64, 0, 8)])
r = pg.NullRenderer(48000); r.set_source(plain8); r.start()
for i in range(8):
    y = plain8.render(i * 48000, 48000).data
    assert float(np.max(np.abs(got4[i].astype(np.float64) - y))) <= 1e-5 * float(np.max(np.abs(y))), i
r.stop()
# random pulls (streams of equal blocks with seeks, steps back, odd lengths, a direct pull of the local mix in between):
# the share through the reducer -- windows reduced whole, rows handed out, windows abandoned half-way -- equals the local
# mix of the owned voices rendered pull by pull (one rank: the sum over ranks is the identity)
from pygmu2_amd import voice_bank
for case in range(6):
    rng = np.random.default_rng(300 + case)
    make, total = ((supersaw_voice, 512), (c4_voice, 64))[case % 2]
    n = int(rng.choice([4096, 12_288, 48_000]))
    pos, pulls = 0, []
    for _ in range(int(rng.integers(14, 26))):
        what = rng.random()
        if what < 0.08:
            pos += int(rng.integers(1, 100_000))
        elif what < 0.12:
            pos = max(0, pos - int(rng.integers(1, 3 * n)))
        size = n if rng.random() < 0.9 else int(rng.choice([17, 5000, 2 * n]))
        pulls.append((pos, size, rng.random() < 0.05))
        pos += size
    shard = ShardedMixPE([make(pg, i) for i in range(total)], 0, 8, reducer=RcclReducer())
    r = pg.NullRenderer(48000); r.set_source(shard); r.start()
    got = []
    for s, m, direct in pulls:
        got.append((shard.local if direct else shard).render(s, m).data.copy())
    r.stop()
    keep_switches = {k: getattr(voice_bank, k) for k in ("BANK_WINDOWS", "LADDER_WINDOWS", "PREFETCH_SUPERSAW_VOICES",
                                                          "PREFETCH_LADDER_INPUT")}
    for k in keep_switches:
        setattr(voice_bank, k, False)
    plain = pg.MixPE(*[make(pg, i) for i in shard_indices(total, 0, 8)])
    r = pg.NullRenderer(48000); r.set_source(plain); r.start()
    want = [plain.render(s, m).data.copy() for s, m, _ in pulls]
    r.stop()
    for k, v in keep_switches.items():
        setattr(voice_bank, k, v)
    peak = max(float(np.max(np.abs(w))) for w in want)
    for i, (a, b) in enumerate(zip(got, want)):
        assert a.shape == b.shape and float(np.max(np.abs(a.astype(np.float64) - b))) <= 1e-6 * peak, (case, i, pulls[i])
comm.destroy()
assert not comm.initialised()
assert "torch" not in sys.modules
print("RCCL_ABI_OK")
'''


OUT_OF_STEP_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PGX_ROOT"])
import pygmu2_amd as pg
from pygmu2_amd import comm, device
from pygmu2_amd.sharding import ShardedMixPE, c5_voice
device.ensure_init()
comm.init(0, 1, comm.unique_id())
pg.set_sample_rate(48000)
root = ShardedMixPE([c5_voice(pg, i) for i in range(6)], 0, 1)
root._world = 2
r = pg.NullRenderer(48000); r.set_source(root); r.start()
try:
    for i in range(6):                       # PGX_COMM_CHECK_FAULT_AT=3: that ticket's check contributes another count
        root.render(i * 4096, 4096).data
except RuntimeError as exc:
    assert "ranks out of step at collective 3" in str(exc), str(exc)
    print("OUT_OF_STEP_SEEN", flush=True)
else:
    raise SystemExit("the planted disagreement went unnoticed")
# the interpreter now ends: the communicator has failed, so the exit hook abandons it and ends the process with 70
'''


def test_ranks_out_of_step_fail_loudly_and_the_process_exits_non_zero(tmp_path):
    """pgx_comm.hip's agreement check (a planted disagreement on a 1-rank communicator) and the exit path of a failed
    communicator: no join / synchronise / ncclCommDestroy that could hang with it, exit status 70."""
    env = dict(os.environ, PGX_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", PGX_COMM_CHECK_FAULT_AT="3")
    script = tmp_path / "w.py"
    script.write_text(OUT_OF_STEP_WORKER)
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert "OUT_OF_STEP_SEEN" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
    assert p.returncode == 70 and "communicator abandoned" in p.stderr, (p.returncode, p.stderr[-2000:])


def _worker(tmp_path, text, token, **env_extra):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PGX_ROOT=ROOT,
               HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    script = tmp_path / "w.py"
    script.write_text(text)
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and token in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_rccl_allreduce_through_the_c_abi(tmp_path):
    _worker(tmp_path, ABI_WORKER, "RCCL_ABI_OK")


def test_rccl_allreduce_on_library_buffers(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PGX_ROOT=ROOT,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_SINGLE_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_bench_distributed_glue_single_rank(tmp_path):
    """bench.py under torch.distributed.run with one rank: same code path as the driver's N>1 launch."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    import json
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["n_ranks_seen"] == 1 and d["value"] > 0 and "voice_mix" in d and "roofline" in d


def test_bench_sharded_workload_line(tmp_path):
    """--workload supersaw as the primary line (what a scaling run reports), one rank."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "supersaw", "--steps", "2",
                        "--warmup", "1", "--no-cpu"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    import json
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["scaling"] == "strong" and d["value"] > 0 and d["mix"]["voices_on_this_rank"] == 512
