"""Frame sharding across the GPUs of one node + the single result gather.

Frames are independent units (the reference handles them one at a time, F4), so a job of F frames is split into
contiguous ranges, one per rank (one process per GPU), with NO collective on the data path.  The only exchange
step is the final variable-length gather of per-frame records (boxes, scores, text) to rank 0:
an all_gather of byte counts followed by an all_gather of one padded uint8 tensor (RCCL over xGMI when the
backend is "nccl"; gloo in the CPU tests).  ~100 B/frame: latency-bound, so it is done once per job/chunk.
"""
import struct

import numpy as np


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


def gather_records(records, device=None, to_all=False):
    """All ranks call; rank 0 gets the concatenation ordered by frame number, other ranks get None (to_all=True: every
    rank gets it — the exchange is an all_gather anyway).  Works without torch.distributed initialised (single process)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sorted(records, key=lambda r: r[0])
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    payload = pack_records(records)
    size = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size)
    mx = int(max(int(s.item()) for s in sizes))
    buf = torch.zeros(mx, dtype=torch.uint8, device=dev)
    buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    bufs = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(bufs, buf)
    if rank != 0 and not to_all:
        return None
    out = []
    for b, s in zip(bufs, sizes):
        out += unpack_records(b[:int(s.item())].cpu().numpy().tobytes())
    return sorted(out, key=lambda r: r[0])
