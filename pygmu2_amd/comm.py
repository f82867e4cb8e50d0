"""
The RCCL communicator behind a sharded MixPE, through the C ABI (pgx_comm_* / pgx_allreduce_* in
include/pygmu_hip.h).  One process per GPU; nothing here imports torch.

    comm.init(rank, world, id_bytes)      every rank, same id (rank 0: comm.unique_id())
    comm.init_from_env(exchange)          RANK / WORLD_SIZE from the launcher, `exchange(id_or_None) -> id`
                                          is the caller's broadcast of 128 bytes from rank 0

The id travels over whatever channel the launcher offers; `file_exchange` is a rendezvous through a
shared directory for launchers that offer none, `torch_exchange` uses an initialised torch.distributed
group (any backend -- gloo is enough: only the 128-byte id goes through it).
"""

from __future__ import annotations

import ctypes as C
import os
import time

from . import device


def unique_id() -> bytes:
    lib = device.load_library()
    n = lib.pgx_comm_unique_id_bytes()
    buf = C.create_string_buffer(n)
    device.check(lib.pgx_comm_unique_id(buf, n), "pgx_comm_unique_id")
    return buf.raw


def init(rank: int, world: int, id_bytes: bytes) -> None:
    lib = device.ensure_init()
    buf = C.create_string_buffer(bytes(id_bytes), len(id_bytes))
    device.check(lib.pgx_comm_init(int(rank), int(world), buf, len(id_bytes)), "pgx_comm_init")


def info() -> tuple[int, int]:
    """(rank, world); world == 0 when no communicator exists."""
    lib = device.load_library()
    r, w = C.c_int(0), C.c_int(0)
    lib.pgx_comm_info(C.byref(r), C.byref(w))
    return r.value, w.value


def initialised() -> bool:
    return info()[1] > 0


def destroy() -> None:
    """Tear the communicator down; RuntimeError if collectives were still outstanding at the deadline
    (PGX_COMM_EXIT_TIMEOUT_MS): the communicator is then abandoned, not destroyed."""
    device.check(device.load_library().pgx_comm_destroy(), "pgx_comm_destroy")


def fold_check(*words) -> None:
    """Facts every rank must share about the collectives that follow (window or block, rows, switches): folded into
    the sequence hash that the library's agreement checks compare across the ranks (pgx_comm_fold_check)."""
    lib = device.load_library()
    for w in words:
        lib.pgx_comm_fold_check(int(w) & 0x7FFFFFFFFFFFFFFF)


def stats() -> dict:
    """Tickets issued, agreement checks completed, the current sequence hash."""
    lib = device.load_library()
    a, b, h = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    lib.pgx_comm_stats(C.byref(a), C.byref(b), C.byref(h))
    return {"issued": a.value, "checks": b.value, "hash": h.value}


def reduce_scalar(value: float, op: str = "sum") -> float:
    """Synchronous sum / max of one float64 over the ranks (bench bookkeeping)."""
    v = C.c_double(float(value))
    device.check(device.ensure_init().pgx_allreduce_scalar_host(C.byref(v), {"sum": 0, "max": 1}[op]),
                 "pgx_allreduce_scalar_host")
    return v.value


def torch_exchange(id_or_none):
    import torch.distributed as dist
    box = [id_or_none]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def file_exchange(directory: str, timeout_s: float = 120.0):
    path = os.path.join(directory, "pgx_comm_id.bin")

    def exchange(id_or_none):
        if id_or_none is not None:
            tmp = path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(id_or_none)
            os.replace(tmp, path)
            return id_or_none
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > timeout_s:
                raise RuntimeError(f"no communicator id appeared at {path}")
            time.sleep(0.01)
        with open(path, "rb") as f:
            return f.read()

    return exchange


def init_from_env(exchange, rank: int | None = None, world: int | None = None) -> tuple[int, int]:
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    device.ensure_init()
    ident = exchange(unique_id() if rank == 0 else None)
    init(rank, world, ident)
    return rank, world
