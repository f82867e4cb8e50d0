"""Worker for test_sharding_gloo.py: one rank of a world_size-N gloo group rendering a
ShardedMixPE over host-side voices (no GPU)."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_voices(pg, n_voices, channels):
    class HostSine(pg.SourcePE):
        def __init__(self, freq):
            self._f = freq

        def channel_count(self):
            return channels

        def is_pure(self):
            return False

        def _render(self, start, duration):
            t = np.arange(start, start + duration, dtype=np.float64) / 48000.0
            y = np.sin(2.0 * np.pi * self._f * t).astype(np.float32).reshape(-1, 1)
            return pg.Snippet(start, np.tile(y, (1, channels)))

    return [HostSine(110.0 * (i + 1)) for i in range(n_voices)]


def host_mixer(pg):
    class HostMix(pg.ProcessingElement):
        def __init__(self, pes):
            self._pes = list(pes)

        def inputs(self):
            return self._pes

        def is_pure(self):
            return True

        def channel_count(self):
            return self._pes[0].channel_count()

        def _render(self, start, duration):
            acc = self._pes[0].render(start, duration).data.copy()
            for pe in self._pes[1:]:
                acc += pe.render(start, duration).data
            return pg.Snippet(start, acc)

    return lambda pes: HostMix(pes)


def run(rank, world, port, n_voices, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist

    import pygmu2_amd as pg
    from pygmu2_amd.sharding import ShardedMixPE, shard_indices

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    pg.set_sample_rate(48000)
    voices = make_voices(pg, n_voices, 2)
    root = ShardedMixPE(voices, rank, world, local_mixer=host_mixer(pg))
    assert [voices.index(v) for v in root.owned] == shard_indices(n_voices, rank, world)
    r = pg.NullRenderer(sample_rate=48000)
    r.set_source(root)
    r.start()
    out_of_step = os.environ.get("PGX_TEST_OUT_OF_STEP", "")
    if out_of_step:
        # rank 1 makes a different pull (another block length, or the same length at another place) at its third
        # collective: every rank must get "ranks out of step" from the agreement check, nobody may hang
        try:
            for i in range(5):
                n, at = 1000, i * 1000
                if rank == 1 and i == 2:
                    n, at = (500, at) if out_of_step == "count" else (1000, at + 7)
                root.render(at, n).data
        except RuntimeError as exc:
            assert "ranks out of step" in str(exc), str(exc)
            with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
                f.write(str(exc))
            # no barrier (the group is not usable for payloads any more) and no teardown of it either: a peer that is
            # already gone makes gloo's own shutdown fail now and then -- the verdict is written, the process ends
            sys.stdout.flush()
            os._exit(0)
        raise SystemExit("the diverging pull went unnoticed")
    blocks = [root.render(i * 1000, 1000).data for i in range(3)]
    assert root._reducer.checks() == 3, "the first collectives are each preceded by an agreement check"
    r.stop()
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.concatenate(blocks))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
