"""GPU: the RCCL leg of ShardedMixPE on a 1-rank "nccl" group -- proves the zero-copy hand-off of a
library DeviceBuffer to torch.distributed (via __cuda_array_interface__), the stream ordering
between the library stream and torch's stream, and that bench.py's multi-GPU glue runs."""

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
    assert d["n_gpus"] == 1 and d["value"] > 0 and "voice_mix" in d and "roofline" in d
