"""N>1 path on CPU: frame sharding + the single variable-length gather, world_size 2 over gloo."""
import os

import numpy as np
import torch.multiprocessing as mp

from vse_amd import parallel


def test_shard_ranges_cover():
    for total in (0, 1, 7, 64, 65, 172800):
        for world in (1, 2, 3, 8):
            r = [parallel.shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_pack_roundtrip():
    recs = [(5, np.arange(16, dtype=np.float32).reshape(2, 4, 2), [("héllo", 0.5), ("字幕", 0.25)]),
            (6, np.zeros((0, 4, 2), np.float32), [])]
    out = parallel.unpack_records(parallel.pack_records(recs))
    assert out[0][0] == 5 and out[0][2] == [("héllo", 0.5), ("字幕", 0.25)] and np.array_equal(out[0][1], recs[0][1])
    assert out[1][0] == 6 and len(out[1][1]) == 0


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = parallel.shard_range(11, rank, world)
    recs = [(f, np.full((f % 3, 4, 2), f, np.float32), [(f"t{f}_{k}", 0.1 * k) for k in range(f % 3)])
            for f in range(lo, hi)]
    out = parallel.gather_records(recs)
    if rank == 0:
        q.put([(r[0], r[1].tolist(), r[2]) for r in out])
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [g[0] for g in got] == list(range(11))
    for f, boxes, texts in got:
        assert len(boxes) == f % 3 and texts == [(f"t{f}_{k}", np.float32(0.1 * k).item()) for k in range(f % 3)]
