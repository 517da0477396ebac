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


def _worker_mode(rank, world, port, q, force_rank, force):
    """One rank only asks for the all_gather form (VSE_GATHER in ITS environment): both ranks must leave gather_mode() with the
    same answer and the exchange must complete."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("VSE_GATHER", None)
    if rank == force_rank and force:
        os.environ["VSE_GATHER"] = force
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mode = parallel.gather_mode()
    assert parallel.gather_mode() == mode                        # cached: decided once per process group
    calls = []
    orig_gather, orig_all = dist.gather, dist.all_gather
    dist.gather = lambda *a, **k: (calls.append("gather"), orig_gather(*a, **k))[1]
    dist.all_gather = lambda *a, **k: (calls.append("all_gather"), orig_all(*a, **k))[1]
    lo, hi = parallel.shard_range(9, rank, world)
    recs = [(f, np.full((1, 4, 2), f, np.float32), [(f"t{f}", 0.5)]) for f in range(lo, hi)]
    out = parallel.gather_records(recs)
    dist.gather, dist.all_gather = orig_gather, orig_all
    q.put((rank, mode, calls, None if out is None else [r[0] for r in out]))
    dist.barrier()
    dist.destroy_process_group()


def _run_mode(force_rank, force):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + (7 if force else 0) + 3 * force_rank
    procs = [ctx.Process(target=_worker_mode, args=(r, 2, port, q, force_rank, force)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))           # a mismatched collective would hang here: the timeout is the test
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return got


def test_gather_mode_is_agreed_collectively_even_when_one_rank_forces_the_fallback():
    """gather vs all_gather is chosen once, by all ranks together (all_reduce(MIN) of static votes), never by catching
    an exception in the middle of the exchange.  Default on gloo: gather (rank 0 alone receives).  The fallback forced on ONE rank
    only — either one — switches BOTH ranks to all_gather, and the exchange completes (no hang, same records)."""
    got = _run_mode(0, None)
    assert [g[1] for g in got] == ["gather", "gather"]
    assert got[0][2] == ["all_gather", "gather"] and got[1][2] == ["all_gather", "gather"]       # sizes, then the payload
    assert got[0][3] == list(range(9)) and got[1][3] is None
    for force_rank in (1, 0):
        got = _run_mode(force_rank, "all_gather")
        assert [g[1] for g in got] == ["all_gather", "all_gather"], got
        assert got[0][2] == ["all_gather", "all_gather"] and got[1][2] == ["all_gather", "all_gather"]
        assert got[0][3] == list(range(9)) and got[1][3] is None


def _worker_bad_override(rank, world, port, q, bad_rank):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("VSE_GATHER", None)
    if rank == bad_rank:
        os.environ["VSE_GATHER"] = "broadcast"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        parallel.gather_mode()
        q.put((rank, "no error"))
    except ValueError as exc:
        q.put((rank, str(exc)))
    dist.barrier()                       # both ranks are still in step: nobody is stuck in the vote's all_reduce
    dist.destroy_process_group()


def test_gather_mode_rejects_an_unknown_override_on_every_rank():
    """an invalid VSE_GATHER on ONE rank must not raise before the collective (the other rank would wait in the
    all_reduce until the backend's timeout): the invalid value is a vote (-1) and BOTH ranks raise behind the all_reduce."""
    assert parallel.gather_mode() == "local"                     # no process group: nothing to decide
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_bad_override, args=(r, 2, port, q, 1)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert "another rank" in got[0] and "'broadcast'" in got[1], got


def _worker_world1_forced(q, port):
    """VSE_FORCE_DIST=1: a world of ONE runs the whole collective sequence (what bench.py does on a one-GPU box with RCCL)."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VSE_FORCE_DIST="1")
    os.environ.pop("VSE_GATHER", None)
    dist.init_process_group("gloo", rank=0, world_size=1)
    assert parallel.dist_active()
    mode = parallel.gather_mode()
    recs = [(f, np.full((1, 4, 2), f, np.float32), [(f"t{f}", 0.5)]) for f in (2, 0, 1)]
    out = parallel.gather_records(recs)
    q.put((mode, [r[0] for r in out], dict(parallel.COLLECTIVES)))
    dist.destroy_process_group()
    parallel.reset_gather_mode()


def test_world_of_one_runs_the_collectives_when_forced():
    assert not parallel.dist_active()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_world1_forced, args=(q, 35500 + (os.getpid() % 2000)))
    p.start()
    mode, frames, calls = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0
    assert mode == "gather" and frames == [0, 1, 2]
    assert calls == {"all_reduce": 1, "gather": 2, "all_gather": 1}, calls     # vote, probe + payload, sizes


# ---- eight ranks on one node: the HOST side of a step under the per-rank thread cap ---------------------------------------
def _host_step(pipe, boxes_per_frame, texts, base):
    """What every rank does on the host for one 64-frame batch between the detector's maps and the gather: box ordering,
    crop geometry, the reference's chunking + the ragged partition, string decode, record packing (the DB hull / rectangle
    geometry itself runs in C++ inside vse_db_postprocess and needs the device part before it)."""
    from vse_amd import pipeline
    ordered = [pipeline.sorted_boxes(b) for b in boxes_per_frame]
    specs = pipe._crop_specs(ordered)
    groups = pipe._groups(specs)
    assert sum(len(g[0]) for g in groups) == len(specs)
    res = [[("".join(pipe.charset[j] for j in texts[(f + k) % len(texts)]), 0.9) for k in range(len(b))] for f, b in enumerate(ordered)]
    recs = [(base + f, np.asarray(b, np.float32).reshape(-1, 4, 2), r) for f, (b, r) in enumerate(zip(ordered, res))]
    return parallel.pack_records(recs), recs


def _host_workload(seed):
    from vse_amd import pipeline
    rng = np.random.default_rng(seed)
    pipe = pipeline.OcrPipeline.__new__(pipeline.OcrPipeline)
    pipe.rec_mode, pipe.rec_h, pipe.rec_base_w, pipe.rec_batch_num = "ragged", 48, 320, 6
    pipe.bucket, pipe.batch_round, pipe.max_rec_batch, pipe.min_rec_group = 256, 4, 64, 8
    pipe.charset = [""] + [chr(0x4e00 + i) for i in range(6000)]
    boxes = []
    for f in range(64):
        bf = []
        for _ in range(int(rng.integers(1, 4))):
            x0, y0 = int(rng.integers(100, 600)), int(rng.integers(700, 950))
            w, h = int(rng.integers(200, 1200)), int(rng.integers(40, 70))
            bf.append(np.array([[x0, y0], [x0 + w, y0], [x0 + w, y0 + h], [x0, y0 + h]], np.float32))
        boxes.append(bf)
    texts = [rng.integers(1, 6000, size=int(rng.integers(5, 25))) for _ in range(37)]
    return pipe, boxes, texts


def _best_host_ms(pipe, boxes, texts, reps=30):
    import time
    best = 1e9
    for r in range(reps):
        t0 = time.perf_counter()
        _host_step(pipe, boxes, texts, r * 64)
        best = min(best, 1e3 * (time.perf_counter() - t0))
    return best


def _worker8(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["WORLD_SIZE"] = str(world)
    cap = parallel.cap_host_threads(world)
    import torch
    torch.set_num_threads(cap)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pipe, boxes, texts = _host_workload(100 + rank)
    _best_host_ms(pipe, boxes, texts, reps=3)              # warm
    dist.barrier()                                         # all eight ranks time their host step AT THE SAME TIME
    ms = _best_host_ms(pipe, boxes, texts)
    _payload, recs = _host_step(pipe, boxes, texts, rank * 64)
    out = parallel.gather_records(recs)
    q.put((rank, ms, cap, torch.get_num_threads(), None if out is None else [r[0] for r in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_host_side_under_the_thread_cap():
    """SURVEY §8(e) first contact with an 8-GPU node, host side: eight gloo ranks run the host part of a step concurrently
    with their thread pools capped at cores / 8, then gather their records to rank 0.  The per-rank host time must stay
    within 1.2 x of a single rank's (+ 0.5 ms of scheduler noise; median rank with a core per rank, every rank with two) — oversubscribed
    thread pools are what would break that — and the gather must return all 8 x 64 records in frame order."""
    cores = os.cpu_count() or 1
    pipe, boxes, texts = _host_workload(100)
    _best_host_ms(pipe, boxes, texts, reps=3)
    one = _best_host_ms(pipe, boxes, texts)
    world = 8
    # A timing bound on a shared machine: up to three attempts, the functional checks on every one of them, the bound on the best
    # (with exactly one core per rank — this build container — the runner, the OS and neighbours take cores from some ranks in
    # about one run of four; the measured figures are printed either way)
    attempts = []
    for attempt in range(3):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 31500 + ((os.getpid() + 97 * attempt) % 2000)
        procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=300) for _ in range(world))
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert all(g[2] == parallel.host_threads_per_rank(world) == g[3] for g in got)         # the cap reached torch
        assert got[0][4] == list(range(world * 64)) and all(g[4] is None for g in got[1:])      # gather to rank 0, frame order
        times = sorted(g[1] for g in got)
        worst = times[-1]
        print(f"host ms per 64-frame step: 1 rank {one:.2f}, 8 concurrent ranks median {times[len(times) // 2]:.2f} worst {worst:.2f} "
              f"({cores} cores, cap {got[0][2]} threads/rank, attempt {attempt + 1})")
        attempts.append(times)
        # the median rank must hold the bound with a core per rank, the worst one only when there are cores to spare (the GPU boxes: 64+)
        ok = cores < world or times[len(times) // 2] <= 1.2 * one + 0.5
        ok = ok and (cores < 2 * world or worst <= 1.2 * one + 0.5)
        if ok:
            break
    assert ok, (one, attempts)
