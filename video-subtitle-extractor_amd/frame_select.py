"""Accurate-mode ("precise") frame selection on top of the batched OCR engine — row a13 of SURVEY §8(a).

The reference decodes a frame, runs the detector on it, maybe runs the full OCR on it, and only then decodes the
next frame (backend/main.py:255-376): three dependent device round trips per frame at batch 1.  Frames are
independent, so here a CHUNK of frames goes through `detect_batch` in one launch sequence (and, optionally, the
frames that show an in-area box through `predict_batch`), after which the start/end-of-subtitle automaton is resolved
on the tiny results on the host.  The emitted OCR tasks are identical to the reference's queue, including its quirks
(tests/golden/frame_loop.json): a start is armed by the first in-area detection and re-armed after each end; the end
of a subtitle is the first frame whose in-area text has Levenshtein ratio <= threshold against the START frame, or the
first frame without an in-area box; cached OCR results older than 10 frames are evicted; when tasks are flushed the
cached result is looked up by the CURRENT frame number (not the queued one), so a task carries boxes/text only when
those coincide.
"""
from collections import deque

from .shim import get_coordinates


def similarity(a, b):
    """Levenshtein.ratio as used at backend/main.py:949: normalised InDel similarity, 1.0 for two empty strings."""
    if not a and not b:
        return 1.0
    la, lb = len(a), len(b)
    row = [0] * (lb + 1)
    for i in range(la):
        diag = 0
        ca = a[i]
        for j in range(lb):
            up = row[j + 1]
            row[j + 1] = diag + 1 if ca == b[j] else (up if up >= row[j] else row[j])
            diag = up
    return 2.0 * row[lb] / (la + lb)


def _inside(c, area):
    return area.xmin <= c[0] and c[1] <= area.xmax and area.ymin <= c[2] and c[3] <= area.ymax


class AccurateFrameSelector:
    # before the first subtitle / waiting for a start / waiting for the end / last frame consumed while tracking
    IDLE, ARMED, TRACKING, DONE = 0, 1, 2, 3

    def __init__(self, detect_batch, predict, sub_area, frame_count, threshold=80, chunk=64, predict_batch=None,
                 detect_stream=None, predict_with_dets=None):
        """detect_batch(list of frames) -> list of ndarray[N,4,2];  predict(frame) -> (boxes, [(text, score)]);
        predict_batch(list of frames) -> list of those (optional: prefetches every frame of a chunk that has an in-area
        box; the automaton asks for a subset of them).
        With an uploader (run): detect_stream(iterable of device batches) -> generator of detect_batch results keeps the
        detector of the next chunks in flight while a chunk is resolved, and predict_with_dets(device frames, their detections)
        -> list of predict results recognises the wanted frames from the boxes the detector already produced (the reference
        runs the same detector a second time inside predict: same frame, same boxes)."""
        self.detect_batch = detect_batch
        self.predict = predict
        self.predict_batch = predict_batch
        self.detect_stream = detect_stream
        self.predict_with_dets = predict_with_dets
        self.area = sub_area
        self.frame_count = frame_count
        self.threshold = threshold / 100.0
        self.chunk = chunk
        self.tasks = []
        self._state = self.IDLE
        self._no = 0
        self._start_no = 0
        self._pending = deque()
        self._ocr = {}                        # frame_no -> {"text","dt_box","rec_res"} (reference's result cache)
        self._prefetched = {}

    # ---- OCR results ------------------------------------------------------------------------------------------
    def _area_text(self, boxes, res):
        if self.area is None:
            return ""                         # the reference appends nothing when no area is set (main.py:914-921)
        return "".join(r[0] for r, c in zip(res, get_coordinates(boxes)) if _inside(c, self.area))

    def _ocr_of(self, no, frame):
        if no in self._prefetched:
            return self._prefetched.pop(no)
        return self.predict(frame)

    def _remember(self, no, frame):
        boxes, res = self._ocr_of(no, frame)
        self._ocr[no] = {"text": self._area_text(boxes, res), "dt_box": boxes, "rec_res": res}

    def _same_as_start(self, frame):
        if self._start_no not in self._ocr:
            self._remember(self._start_no, None)      # cannot happen in the reference either unless evicted mid-run
        if self._no not in self._ocr:
            self._remember(self._no, frame)
        a, b = self._ocr[self._start_no]["text"], self._ocr[self._no]["text"]
        horizon = min(self._start_no, self._no) - 10
        for k in [k for k in self._ocr if k < horizon]:
            del self._ocr[k]
        return similarity(a, b) > self.threshold

    # ---- task queue ---------------------------------------------------------------------------------------------
    def _flush(self, keep):
        while len(self._pending) > keep:
            no = self._pending.popleft()
            hit = self._ocr.get(self._no)
            self.tasks.append((self.frame_count, no, hit["dt_box"], hit["rec_res"]) if hit else
                              (self.frame_count, no, None, None))

    # ---- automaton ------------------------------------------------------------------------------------------------
    def _has_subtitle(self, boxes):
        if self.area is None:
            return len(boxes) > 0
        return any(_inside(c, self.area) for c in get_coordinates(boxes.tolist()))

    def _step(self, frame, boxes):
        self._no += 1
        has = self._has_subtitle(boxes)
        if has and self._state == self.IDLE and self.area is not None:
            self._state = self.ARMED
        if has:
            if self._state == self.ARMED:
                self._start_no = self._no
                had = self._no in self._ocr
                b, r = self._ocr_of(self._no, frame)
                if not had:
                    self._ocr[self._no] = {"text": self._area_text(b, r), "dt_box": b, "rec_res": r}
                    self._pending.append(self._no)
                self._state = self.TRACKING
            if self._state == self.TRACKING and self._no == self.frame_count:
                self._state = self.DONE
                self._pending.append(self._no)
            if self._state == self.TRACKING and not self._same_as_start(frame):
                self._state = self.ARMED
                self._pending.append(self._no - 1)
        elif self._state == self.TRACKING:
            self._state = self.ARMED
            self._pending.append(self._no - 1)
        self._flush(1)

    def run(self, frames, uploader=None):
        """frames: iterable of frames in decode order.  Returns the task list
        [(frame_count, frame_no, dt_box | None, rec_res | None)] in the reference's queue order.
        uploader (staging.Uploader): chunks are staged through pinned memory by a producer thread; detect_batch and
        predict_batch then receive device uint8 tensors [n,H,W,3] (one upload per chunk serves both)."""
        def chunks():
            buf = []
            for f in frames:
                if buf and (len(buf) == self.chunk or getattr(buf[0][1], "shape", None) != getattr(f, "shape", None)):
                    yield buf
                    buf = []
                buf.append((None, f))
            if buf:
                yield buf
        if uploader is not None:
            from . import staging
            staged_chunks = staging.prefetch(chunks(), uploader)
            if self.detect_stream is not None:
                pending = deque()

                def tensors():
                    for items, staged in staged_chunks:
                        dev = staged.tensor()
                        pending.append((items, dev))
                        yield dev
                for dets in self.detect_stream(tensors()):
                    items, dev = pending.popleft()
                    self._consume([f for _, f in items], dev, dets)
            else:
                for items, staged in staged_chunks:
                    self._consume([f for _, f in items], staged.tensor())
        else:
            for items in chunks():
                self._consume([f for _, f in items])
        self._flush(0)
        return self.tasks

    def _consume(self, frames, dev=None, dets=None):
        if dets is None:
            dets = self.detect_batch(frames if dev is None else dev)
        if self.predict_batch is not None or (dev is not None and self.predict_with_dets is not None):
            want = [i for i, b in enumerate(dets) if self._has_subtitle(b)]
            if want:
                sub = [frames[i] for i in want] if dev is None else dev[want]
                if dev is not None and self.predict_with_dets is not None:
                    got = self.predict_with_dets(sub, [dets[i] for i in want])
                else:
                    got = self.predict_batch(sub)
                for i, r in zip(want, got):
                    self._prefetched[self._no + 1 + i] = r
        for f, b in zip(frames, dets):
            self._step(f, b)
        self._prefetched.clear()
