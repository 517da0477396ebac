"""Frames -> raw.txt -> SRT: the reference's SubtitleExtractor.run (backend/main.py:103-191) around the batched OCR engine.

What the reference does with one process, two threads and bounded queues — pick the frames to look at (fps sampler
`extract_frame_by_fps`, main.py:228-253, or the accurate-mode detector loop, main.py:255-376), seek + read + crop each of
them (`ocr_task_producer` / `frame_preprocess`, backend/tools/subtitle_ocr.py:163-208,270-289), OCR it and filter its lines
(`ocr_task_consumer` -> `extract_subtitles`, subtitle_ocr.py:20-85,126-161), clean raw.txt (main.py:506-612,671-729) and write
the SRT (main.py:614-637) — is done here as: task list -> BATCHES of frames through `predict_batch` -> the same per-frame line
logic on the host.  Frames are independent, so the tasks of one video can also be sharded over ranks (`shard=(rank, world)`)
and the per-frame records gathered on rank 0 (parallel.gather_records) before the sequential text logic runs.

Pinned by tests/golden/extract.json (the reference's own producer / consumer / fps sampler executed on scripted inputs),
frame_loop.json, srt.json, raw_filters.json and text_cleanup.json.  Not rebuilt: VideoSubFinder frame selection (closed binary), GUI
progress plumbing; reformat.execute is text_cleanup.py with the word segmenter as a parameter (its corpus is not installed here).

Frame sources: anything with `frame_count`, `fps`, `read(frame_no) -> uint8 BGR [H,W,3] | None` (1-based, like
cap.set(CAP_PROP_POS_FRAMES, frame_no - 1); cap.read()) and `frames()` (decode order).  `ArraySource` wraps decoded frames;
ingest.py reads the lossless containers that need no codec (neither box has cv2 / ffmpeg).
"""
import re
from collections import deque
from types import SimpleNamespace

import numpy as np

from . import frame_select, parallel, raw_filters, shim, srt, staging, text_cleanup

# backend/tools/constant.py:5-13 — default subtitle position used by the fps sampler's half-frame crop
LOWER_PART, UPPER_PART, UNKNOWN = "LOWER_PART", "UPPER_PART", "UNKNOWN"


class SubtitleArea(SimpleNamespace):
    """backend/bean/subtitle_area.py: (ymin, ymax, xmin, xmax) in frame pixels."""

    def __init__(self, ymin, ymax, xmin, xmax):
        super().__init__(ymin=ymin, ymax=ymax, xmin=xmin, xmax=xmax)


class ArraySource:
    def __init__(self, frames, fps):
        self._frames = frames
        self.frame_count = len(frames)
        self.fps = float(fps)

    def read(self, frame_no):
        return self._frames[frame_no - 1] if 1 <= frame_no <= self.frame_count else None

    def frames(self):
        return iter(self._frames)

    pos_msec = None        # no container timestamps: SRT time codes fall back to frame_no / fps (main.py:745-748)


def frame_preprocess(subtitle_area, frame):
    """Half-frame crop of subtitle_ocr.py:270-289 (a view, like the reference's slice)."""
    if subtitle_area == LOWER_PART:
        return frame[int(frame.shape[0] // 2):]
    if subtitle_area == UPPER_PART:
        return frame[:int(frame.shape[0] // 2)]
    return frame


def fps_tasks(frame_count, fps, extract_frequency, default_area=None):
    """extract_frame_by_fps (main.py:228-253): one task per read that is followed by int(fps // frequency) - 1 skipped reads.
    Task = (total_frame_count, frame_no, dt_box, rec_res, total_ms, default_subtitle_area)."""
    tasks = []
    reads = no = 0
    skip = int(fps // extract_frequency) - 1
    while reads < frame_count:
        reads += 1
        no += 1
        tasks.append((frame_count, no, None, None, None, default_area))
        for _ in range(skip):
            if reads < frame_count:
                reads += 1
                no += 1
    return tasks


def frame_lines(frame_no, dt_box, rec_res, sub_area, rec_char_type, drop_score, deviation_rate):
    """extract_subtitles (subtitle_ocr.py:20-85) for one frame: raw.txt lines of the recognised text that passes the filters."""
    return shim.extract_subtitles(frame_no, (dt_box, rec_res), sub_area, rec_char_type, deviation_rate, drop_score)


def run_ocr_tasks(source, tasks, ocr, sub_area=None, rec_char_type="ch", drop_score=0.75, deviation_rate=0.0, batch=64,
                  shard=None, gather_device=None, uploader=None):
    """Producer + consumer of subtitle_ocr.py over a task list.  `ocr` has predict(frame) and optionally
    predict_batch(batch of equal-shaped frames).  Tasks whose frame cannot be read are skipped like the reference's failed
    cap.read(); tasks that carry a cached (dt_box, rec_res) — accurate mode — are not recognised again.
    uploader (staging.Uploader): batches are assembled in pinned memory and uploaded by a producer thread while the previous
    batch is recognised — the reference's producer / consumer pair at batch granularity; predict_batch then receives a
    device uint8 tensor [n,H,W,3].  Without one the frames are stacked on the host.
    shard=(rank, world): this rank recognises a contiguous slice of the tasks; every rank gets the records of all tasks back
    (one variable-length gather) and therefore returns the same lines.  -> list of raw.txt lines in task order."""
    tasks = [t for t in tasks if t[1] != -1]
    lo, hi = (0, len(tasks)) if shard is None else parallel.shard_range(len(tasks), *shard)
    if shard is not None and shard[1] > 1:
        parallel.cap_host_threads(shard[1])       # one process per GPU: this rank's numpy / torch pools get cores / world threads
    results = {}                        # task index -> (dt_box, rec_res)
    batched = hasattr(ocr, "predict_batch")

    def batches():
        """lists of (task index, frame): consecutive readable tasks of one frame shape, at most `batch` of them"""
        pend = []
        for k in range(lo, hi):
            _total, no, dt_box, rec_res, _ms, default_area = tasks[k]
            frame = source.read(no)
            if frame is None:
                continue
            if dt_box is not None and rec_res is not None:
                results[k] = (dt_box, rec_res)
                continue
            if default_area is not None:
                frame = frame_preprocess(default_area, frame)
            if pend and (pend[0][1].shape != frame.shape or len(pend) >= batch):
                yield pend
                pend = []
            pend.append((k, frame))
        if pend:
            yield pend

    if batched and uploader is not None:
        staged = staging.prefetch(batches(), uploader)
        if hasattr(ocr, "predict_stream"):
            # the recogniser overlaps the detector of the next batches with the recognition of the current one; results come
            # back in batch order, so the item lists are matched through a queue
            pending = deque()

            def tensors():
                for items, sb in staged:
                    pending.append(items)
                    yield sb.tensor()
            for out in ocr.predict_stream(tensors()):
                for (k, _), r in zip(pending.popleft(), out):
                    results[k] = r
        else:
            for items, sb in staged:
                for (k, _), r in zip(items, ocr.predict_batch(sb.tensor())):
                    results[k] = r
    else:
        for items in batches():
            frames = [f for _, f in items]
            out = ocr.predict_batch(_stack(frames)) if batched and len(frames) > 1 else [ocr.predict(f) for f in frames]
            for (k, _), r in zip(items, out):
                results[k] = r
    if shard is not None and shard[1] > 1:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() != shard[1]:
            raise RuntimeError(f"run_ocr_tasks(shard={shard}): torch.distributed is not initialised with world size {shard[1]} — "
                               "this rank recognised only its slice of the tasks and the other slices cannot be gathered")
        recs = [(k, _boxes_array(results[k][0]), list(results[k][1])) for k in sorted(results)]
        results = {k: (_boxes_list(b), r) for k, b, r in parallel.gather_records(recs, device=gather_device, to_all=True)}
    lines = []
    for k in sorted(results):
        dt_box, rec_res = results[k]
        lines += frame_lines(tasks[k][1], dt_box, rec_res, sub_area, rec_char_type, drop_score, deviation_rate)
    return lines


def _stack(frames):
    try:
        import torch
        if isinstance(frames[0], torch.Tensor):
            return torch.stack(frames)
        return torch.from_numpy(np.stack(frames)).to(shim._context().tdev)
    except ImportError:                 # host-only use with a scripted recogniser (tests)
        return np.stack(frames)


def _boxes_array(dt_box):
    return np.asarray(dt_box, np.float32).reshape(-1, 4, 2)


def _boxes_list(arr):
    # OcrRecogniser.predict returns lists of four (x, y) int tuples (ocr.py:80-82); get_coordinates needs a list
    return [[(int(x), int(y)) for x, y in q] for q in np.asarray(arr).reshape(-1, 4, 2)]


class SubtitleExtractor:
    """run() = backend/main.py:103-191 without the GUI/process plumbing.

    mode 'accurate' (+ a subtitle area): frames are chosen by the detector loop (frame_select.AccurateFrameSelector);
    otherwise by the fps sampler (the reference would use the closed VideoSubFinder binary in fast/auto mode when an area is
    given; the sampler is what it runs without an area and on platforms without that binary)."""

    def __init__(self, source, ocr, detect_batch=None, sub_area=None, mode="fast", language="ch", extract_frequency=3,
                 default_subtitle_area=None, drop_score=0.75, deviation_rate=0.0, threshold=80, batch=64,
                 watermark_decide=None, scene_text_decide=lambda band: True, shard=None, gather_device=None,
                 word_segmentation=False, segment=None, uploader=None, detect_stream=None):
        self.source, self.ocr, self.detect_batch = source, ocr, detect_batch
        self.sub_area, self.mode, self.language = sub_area, mode, language
        self.extract_frequency, self.default_subtitle_area = extract_frequency, default_subtitle_area
        self.drop_score, self.deviation_rate, self.threshold, self.batch = drop_score, deviation_rate, threshold, batch
        self.watermark_decide, self.scene_text_decide = watermark_decide, scene_text_decide
        self.shard, self.gather_device = shard, gather_device
        # staging.Uploader (or "auto": one on the shim's device when there is a GPU): detect_batch / predict_batch then receive
        # device uint8 tensors [n,H,W,3] staged through pinned memory by a producer thread instead of lists of host frames
        self.uploader = uploader
        # accurate mode with an uploader: detect_stream(iterable of device batches) -> generator of detect_batch results (the
        # detector of the next chunks stays in flight); ocr.predict_with_dets, when it exists, recognises from those boxes
        self.detect_stream = detect_stream
        self.word_segmentation, self.segment = word_segmentation, segment      # config.wordSegmentation (main.py:181-182)
        self.raw_lines = None
        self.short_lines = None

    def _uploader(self):
        if self.uploader == "auto":
            self.uploader = staging.default_uploader() if hasattr(self.ocr, "predict_batch") else None
        return self.uploader

    def select_tasks(self):
        s = self.source
        if self.sub_area is not None and self.mode == "accurate" and self.detect_batch is not None:
            up = self._uploader()
            sel = frame_select.AccurateFrameSelector(self.detect_batch, self.ocr.predict, self.sub_area, s.frame_count,
                                                     self.threshold, chunk=self.batch,
                                                     predict_batch=getattr(self.ocr, "predict_batch", None) and self._predict_list,
                                                     detect_stream=self.detect_stream,
                                                     predict_with_dets=getattr(self.ocr, "predict_with_dets", None))
            return [(t[0], t[1], t[2], t[3], None, None) for t in sel.run(s.frames(), uploader=up)]
        return fps_tasks(s.frame_count, s.fps, self.extract_frequency, self.default_subtitle_area)

    def _predict_list(self, frames):
        return self.ocr.predict_batch(frames if not isinstance(frames, list) else _stack(frames))

    def run(self):
        """-> SRT text.  raw_lines (normalised, as the reference rewrites raw.txt) and short_lines are kept on the object."""
        tasks = self.select_tasks()
        lines = run_ocr_tasks(self.source, tasks, self.ocr, self.sub_area, self.language, self.drop_score,
                              self.deviation_rate, self.batch, self.shard, self.gather_device, self._uploader())
        if self.sub_area is None:
            if self.watermark_decide is not None:               # the reference asks on stdin (main.py:164-170)
                lines = raw_filters.filter_watermark(lines, self.watermark_decide)
            lines = raw_filters.filter_scene_text(lines, self.scene_text_decide) if lines else lines
        text, self.short_lines, self.raw_lines = srt.generate_subtitle_file(lines, self.source.fps, self.threshold,
                                                                              getattr(self.source, "pos_msec", None))
        if self.word_segmentation:
            text, _ = text_cleanup.cleanup_srt(text, self.language, self.segment or text_cleanup.default_segmenter())
        return text

    @staticmethod
    def srt2txt(srt_text):
        """main.py:1037-1043 (pysrt: every block's text, one block after the other)."""
        blocks = [b for b in re.split(r"\n(?=\d+\n\d\d:\d\d:\d\d,\d{3} --> )", srt_text) if b.strip()]
        return "".join(b.split("\n", 2)[2].rstrip("\n") + "\n" for b in blocks)
