"""Host -> HBM frame staging for the callers of the hot path (extractor.py, frame_select.py).

The reference hands one decoded frame at a time to paddle, which copies it to the device itself (ocr.py:27); its producer
thread (subtitle_ocr.py:163-208) only keeps the decoder busy while the consumer recognises.  A batched engine needs the
same overlap at batch granularity, and a 1080p batch is 400 MB: `np.stack` + a pageable copy costs more than the detector +
recogniser of the mobile models.  Here a batch is assembled ONCE, frame by frame, in a pinned slab (a small thread pool: numpy
releases the GIL while it copies), sent with one asynchronous copy on a copy stream, and the next batch is staged by a
producer thread while the current one is recognised (`prefetch`).

    up = Uploader(device)                         # torch.device / "cuda:0"
    for items, batch in prefetch(batches, up):    # batches: iterable of lists of (key, frame) with equal frame shapes
        dev = batch.tensor()                      # uint8 [n,H,W,3] on the device, ordered after the copy on the current stream
"""
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np


class StagedBatch:
    def __init__(self, dev, event):
        self._dev, self._event = dev, event

    def tensor(self):
        import torch
        cur = torch.cuda.current_stream(self._dev.device)
        cur.wait_event(self._event)
        self._dev.record_stream(cur)
        return self._dev


class Uploader:
    def __init__(self, device, depth=3, workers=4):
        import torch
        self.device = torch.device(device)
        self.depth = depth
        self._slabs = [None] * depth            # pinned uint8 buffers, grown on demand
        self._busy = [None] * depth             # event of the last copy out of each slab
        self._k = 0
        self._stream = torch.cuda.Stream(device=self.device)
        self._pool = ThreadPoolExecutor(workers)
        self._lock = threading.Lock()

    def _slab(self, nbytes):
        import torch
        with self._lock:
            k = self._k
            self._k = (k + 1) % self.depth
        if self._busy[k] is not None:
            self._busy[k].synchronize()         # the copy that last read this slab has finished
        if self._slabs[k] is None or self._slabs[k].numel() < nbytes:
            self._slabs[k] = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        return k, self._slabs[k]

    def stage(self, frames):
        """list of equal-shaped uint8 frames (views are fine) -> StagedBatch"""
        import torch
        shape = tuple(frames[0].shape)
        per = int(np.prod(shape))
        k, slab = self._slab(per * len(frames))
        host = slab[:per * len(frames)].view(len(frames), *shape)
        dst = host.numpy()
        list(self._pool.map(lambda i: np.copyto(dst[i], frames[i]), range(len(frames))))
        with torch.cuda.stream(self._stream):
            dev = torch.empty((len(frames),) + shape, dtype=torch.uint8, device=self.device)
            dev.copy_(host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        self._busy[k] = ev
        return StagedBatch(dev, ev)

    def bind_thread(self):
        """called once by a thread that is going to stage batches"""
        import torch
        torch.cuda.set_device(self.device)

    def close(self):
        self._pool.shutdown(wait=False)


def default_uploader():
    """An Uploader on the shim's device when there is a GPU, else None (callers then stack frames on the host)."""
    try:
        import torch
    except ImportError:
        return None
    if not torch.cuda.is_available():
        return None
    from . import shim
    return Uploader(shim._context().tdev)


_END = object()


def prefetch(batches, uploader, ahead=2):
    """Iterate (items, StagedBatch) while a producer thread reads + stages up to `ahead` batches in advance.  `batches` yields
    lists of (key, frame); an exception of the producer (a failing decoder) is re-raised in the consumer."""
    q = queue.Queue(maxsize=max(1, ahead))
    stop = threading.Event()

    def produce():
        try:
            uploader.bind_thread()
            for items in batches:
                if stop.is_set():
                    return
                q.put((items, uploader.stage([f for _, f in items])))
            q.put(_END)
        except BaseException as e:             # noqa: BLE001 - handed to the consumer
            q.put(e)

    th = threading.Thread(target=produce, daemon=True)
    th.start()
    try:
        while True:
            got = q.get()
            if got is _END:
                break
            if isinstance(got, BaseException):
                raise got
            yield got
    finally:
        stop.set()
        while th.is_alive():                   # unblock a producer waiting on a full queue
            try:
                q.get_nowait()
            except queue.Empty:
                th.join(0.01)
