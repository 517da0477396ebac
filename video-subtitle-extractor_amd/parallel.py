"""Frame sharding across the GPUs of one node + the single result gather.

Frames are independent units (the reference handles them one at a time, F4), so a job of F frames is split into
contiguous ranges, one per rank (one process per GPU), with NO collective on the data path.  The only exchange
step is the final variable-length gather of per-frame records (boxes, scores, text) to rank 0:
an all_gather of byte counts (8 B per rank: every rank needs the padded size) followed by a gather of one padded uint8
tensor to rank 0 (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).  ~100 B/frame: latency-bound, so it is done once per job/chunk.
Whether the backend's `gather` or the `all_gather` form is used is agreed by all ranks once, at start-up (gather_mode).
"""
import os
import struct

import numpy as np


def host_threads_per_rank(world, cores=None):
    """Host threads one rank may use when `world` ranks share a node: cores / world, at least 1."""
    cores = cores if cores is not None else (os.cpu_count() or 1)
    return max(1, cores // max(1, world))


def cap_host_threads(world=None, cores=None):
    """One process per GPU: every rank runs the host side of the path (DB geometry, crop grouping, record packing, string
    decode) — numpy / torch / OpenMP pools default to ALL cores each, so 8 ranks would run 8 x cores threads.  Caps this
    process's pools at cores / world.  The environment variables only reach libraries that have not started their pools yet
    (bench.py calls this before importing torch); torch's own pools are set directly.  Returns the cap (None when world <= 1)."""
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
    if world <= 1:
        return None
    n = host_threads_per_rank(world, cores)
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ[var] = str(n)
    import sys
    if "torch" in sys.modules:
        import torch
        torch.set_num_threads(n)
        try:
            torch.set_num_interop_threads(max(1, min(n, 4)))
        except RuntimeError:          # already started: the intra-op cap above is the one that matters
            pass
    return n


def shard_range(total, rank, world):
    """Contiguous [lo, hi) frame range of `rank` (ranges differ by at most one frame)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_records(records):
    """records: list of (frame_no:int, boxes ndarray[k,4,2] float32, texts list[(str,float)]) -> bytes."""
    out = bytearray()
    out += struct.pack("<i", len(records))
    for frame_no, boxes, texts in records:
        b = np.asarray(boxes, dtype=np.float32).reshape(-1, 4, 2)
        assert len(b) == len(texts)
        out += struct.pack("<ii", int(frame_no), len(b))
        out += b.tobytes()
        for t, s in texts:
            raw = t.encode("utf-8")
            out += struct.pack("<fi", float(s), len(raw))
            out += raw
    return bytes(out)


def unpack_records(buf):
    mv = memoryview(buf)
    (n,) = struct.unpack_from("<i", mv, 0)
    pos = 4
    recs = []
    for _ in range(n):
        frame_no, k = struct.unpack_from("<ii", mv, pos)
        pos += 8
        boxes = np.frombuffer(mv, dtype=np.float32, count=k * 8, offset=pos).reshape(k, 4, 2).copy()
        pos += k * 32
        texts = []
        for _ in range(k):
            s, ln = struct.unpack_from("<fi", mv, pos)
            pos += 8
            texts.append((bytes(mv[pos:pos + ln]).decode("utf-8"), s))
            pos += ln
        recs.append((frame_no, boxes, texts))
    return recs


_GATHER_MODE = {}          # token of the default process group -> "gather" | "all_gather"
COLLECTIVES = {}           # name -> calls issued by this module / counted by its callers (bench.py reports them)


def _count(name):
    COLLECTIVES[name] = COLLECTIVES.get(name, 0) + 1


def dist_active():
    """Is there an exchange step to run?  Yes with more than one rank — and with ONE rank when VSE_FORCE_DIST=1: the whole
    collective sequence (all_reduce vote, probe gather, size all_gather, payload gather, barrier) then runs on a world of one, which
    is how a one-GPU box executes the RCCL code path the 8-GPU job takes (tests/test_gpu_bench.py)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("VSE_FORCE_DIST", "0") == "1"


def _group_token():
    """Identifies THIS initialisation of the default process group: the group object carries the token, so a group destroyed and
    re-created in one process (whose id() may be reused) never inherits the previous group's cached decision."""
    import torch.distributed as dist
    g = dist.group.WORLD
    tok = getattr(g, "_vse_token", None)
    if tok is None:
        tok = object()
        try:
            g._vse_token = tok
        except AttributeError:            # a group type without a __dict__: fall back to (id, backend, world), cleared by reset()
            return (id(g), dist.get_backend(), dist.get_world_size())
    return tok


def reset_gather_mode():
    """Forget every cached decision (call next to dist.destroy_process_group())."""
    _GATHER_MODE.clear()


def gather_mode(device=None):
    """How the record exchange reaches rank 0 — `dist.gather` (the other ranks receive nothing) or `dist.all_gather` (every rank
    receives world x the payload and drops it) — decided ONCE per process group, COLLECTIVELY, before any timed region:
    every rank votes from static knowledge only (the backend's name; VSE_GATHER=gather|all_gather overrides a rank's vote), the
    votes are combined with all_reduce(MIN) — a collective every backend has — so all ranks leave with the same answer even when
    only one of them asked for the fallback.  An INVALID override is a vote too (-1): it travels through the same all_reduce and
    every rank raises afterwards — a rank that raised before the collective would leave the others waiting in it.  When the answer
    is "gather" the ranks then run one 8-byte gather as a start-up check; an error there PROPAGATES (with the remedy in its
    message).  Nothing is ever decided by catching an exception in the middle of the exchange: a rank-local failure there would
    leave the ranks in different collectives."""
    import torch
    import torch.distributed as dist
    if not dist_active():
        return "local"
    backend, world, rank = dist.get_backend(), dist.get_world_size(), dist.get_rank()
    key = _group_token()
    if key in _GATHER_MODE:
        return _GATHER_MODE[key]
    dev = device if device is not None else ("cuda" if backend == "nccl" else "cpu")
    want = os.environ.get("VSE_GATHER", "").strip().lower()
    if want not in ("", "gather", "all_gather"):
        vote = -1
    else:
        vote = 0 if want == "all_gather" else (1 if want == "gather" or backend in ("nccl", "gloo") else 0)
    v = torch.tensor([vote], dtype=torch.int32, device=dev)
    dist.all_reduce(v, op=dist.ReduceOp.MIN)
    _count("all_reduce")
    agreed = int(v.item())
    if agreed < 0:
        raise ValueError("VSE_GATHER: expected 'gather' or 'all_gather'" + (f", this rank has {want!r}" if vote < 0 else
                         " (another rank's value is invalid)"))
    mode = "gather" if agreed == 1 else "all_gather"
    if mode == "gather":
        probe = torch.full((8,), rank, dtype=torch.uint8, device=dev)
        got = [torch.zeros(8, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
        try:
            dist.gather(probe, got, dst=0)
            _count("gather")
        except Exception as exc:
            raise RuntimeError(f"the start-up probe of dist.gather failed on rank {rank} (backend {backend}): set VSE_GATHER=all_gather on "
                               f"every rank to use the all_gather form of the record exchange") from exc
        if rank == 0:
            assert [int(g[0].item()) for g in got] == list(range(world)), "gather probe returned the wrong ranks' bytes"
    _GATHER_MODE[key] = mode
    return mode


def gather_records(records, device=None, to_all=False):
    """All ranks call; rank 0 gets the concatenation ordered by frame number, other ranks get None (to_all=True: every
    rank gets it through an all_gather instead of the gather).  Works without torch.distributed initialised (single process).
    gather vs all_gather: gather_mode() (decided once per process group; call it before a timed region — bench.py does)."""
    import torch
    import torch.distributed as dist
    if not dist_active():
        return sorted(records, key=lambda r: r[0])
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    mode = "all_gather" if to_all else gather_mode(dev)
    payload = pack_records(records)
    size = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size)
    _count("all_gather")
    mx = int(max(int(s.item()) for s in sizes))
    buf = torch.zeros(mx, dtype=torch.uint8, device=dev)
    buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    if mode == "all_gather":
        bufs = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(bufs, buf)
        _count("all_gather")
    else:
        # north_star: "RCCL ... only for the final box/text gather" — a gather to rank 0: the other ranks send their padded
        # buffer once and receive nothing (an all_gather would deliver world x the payload to ranks that drop it).  Errors propagate.
        bufs = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, bufs, dst=0)
        _count("gather")
    if rank != 0 and not to_all:
        return None
    out = []
    for b, s in zip(bufs, sizes):
        out += unpack_records(b[:int(s.item())].cpu().numpy().tobytes())
    return sorted(out, key=lambda r: r[0])
