#!/usr/bin/env python3
"""Generates tests/golden/extract.json by RUNNING the reference's own frame-to-raw.txt path in this container on scripted
inputs:

  * SubtitleExtractor.extract_frame_by_fps (backend/main.py:228-253): which frames become OCR tasks for a given frame
    count / frame rate / extractFrequency;
  * backend/tools/subtitle_ocr.py: subtitle_extract_handler = ocr_task_producer (seek + read + frame_preprocess) and
    ocr_task_consumer -> extract_subtitles (get_coordinates, the `en` CJK strip, the subtitle-area / deviation / confidence
    filter, the raw.txt line format) — the reference's code, statement for statement.

None of cv2, shapely, paddleocr, PIL fonts, tqdm, qfluentwidgets is installed: they are replaced by stubs.  Stubs that carry
behaviour, and are therefore part of what the vectors pin:
  * cv2.VideoCapture over an in-memory list of frames (set(CAP_PROP_POS_FRAMES, n) + read() like a seekable file);
  * shapely.geometry.Polygon for AXIS-ALIGNED RECTANGLES only (everything this path builds): area and intersection exact;
  * OcrRecogniser.predict -> scripted (boxes, [(text, score)]) per frame; the frame's number is encoded in its pixels, the
    shape the recogniser saw is recorded (pins the half-frame crop of frame_preprocess).
Only inputs and outputs are written (data, not source).  Needs /root/reference; not run on the GPU box.
"""
import importlib.util
import json
import os
import queue
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "extract.json")
H, W = 40, 64


class _Val:
    def __init__(self, v):
        self.value = v


class RectPolygon:
    """shapely.geometry.Polygon restricted to axis-aligned rectangles given as 4 corners."""

    def __init__(self, pts=None, box=None):
        if box is None:
            xs, ys = [p[0] for p in pts], [p[1] for p in pts]
            assert sorted(set(xs)).__len__() <= 2 and sorted(set(ys)).__len__() <= 2, "rectangles only"
            box = (min(xs), min(ys), max(xs), max(ys))
        self.box = box

    @property
    def area(self):
        x0, y0, x1, y1 = self.box
        return max(0, x1 - x0) * max(0, y1 - y0)

    @property
    def is_empty(self):
        # shapely: two rectangles that only touch intersect in a line / point (not empty); disjoint -> empty
        return self.box is None

    def intersection(self, o):
        x0, y0 = max(self.box[0], o.box[0]), max(self.box[1], o.box[1])
        x1, y1 = min(self.box[2], o.box[2]), min(self.box[3], o.box[3])
        if x0 > x1 or y0 > y1:
            e = RectPolygon(box=(0, 0, 0, 0))
            e.box = None
            return e
        return RectPolygon(box=(x0, y0, x1, y1))


class _Fmt(str):
    def format(self, *a, **k):          # the UI strings take 1-3 arguments; their text is not part of what is pinned
        return ""


class _Tr(dict):
    def __missing__(self, k):
        return _Tr() if k == "Main" else _Fmt()


def frame_of(no, h=H, w=W):
    f = np.zeros((h, w, 3), np.uint8)
    f[:, :, 0] = no & 255
    f[:, :, 1] = no >> 8
    return f


class FakeCapture:
    frames = []

    def __init__(self, path=None):
        self.pos = 0
        self.open = True

    def isOpened(self):
        return self.open

    def set(self, prop, v):
        assert prop == 1, "only CAP_PROP_POS_FRAMES is scripted"
        self.pos = int(v)

    def read(self):
        if self.pos < 0 or self.pos >= len(FakeCapture.frames):
            return False, None
        self.pos += 1
        return True, FakeCapture.frames[self.pos - 1].copy()

    def get(self, prop):
        return 0

    def release(self):
        self.open = False


def mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def install_common():
    class _Tqdm:
        total, n = 0, 0

        def __init__(self, *a, total=0, **k):
            self.total, self.n = total, 0

        def update(self, n):
            self.n += n

        @staticmethod
        def write(*a, **k):
            pass
    mod("cv2", VideoCapture=FakeCapture, CAP_PROP_POS_MSEC=0, CAP_PROP_POS_FRAMES=1, rectangle=lambda *a, **k: a[0],
        imwrite=lambda *a, **k: True)
    mod("tqdm", tqdm=_Tqdm)
    mod("shapely")
    mod("shapely.geometry", Polygon=RectPolygon)
    mod("paddle")
    mod("paddleocr", PaddleOCR=object)
    pil = mod("PIL")
    pil.ImageFont = types.SimpleNamespace(truetype=lambda *a, **k: None)
    pil.ImageDraw = types.SimpleNamespace()
    pil.Image = types.SimpleNamespace()
    for n in ("ImageFont", "ImageDraw", "Image"):
        sys.modules["PIL." + n] = getattr(pil, n)
    pkg = mod("backend")
    pkg.__path__ = [os.path.join(REF, "backend")]
    tools = mod("backend.tools")
    tools.__path__ = [os.path.join(REF, "backend", "tools")]
    bean = mod("backend.bean")
    bean.__path__ = [os.path.join(REF, "backend", "bean")]
    mod("backend.config", config=types.SimpleNamespace(), tr=_Tr(), BASE_DIR=REF)

    class _HA:
        onnx_providers = []

        @classmethod
        def instance(cls):
            return cls()

        def has_cuda(self):
            return False
    mod("backend.tools.hardware_accelerator", HardwareAccelerator=_HA)
    mod("backend.tools.paddle_model_config", PaddleModelConfig=object)
    load("backend.tools.constant", "backend/tools/constant.py")
    load("backend.bean.subtitle_area", "backend/bean/subtitle_area.py")
    load("backend.tools.ocr", "backend/tools/ocr.py")


def quad(x0, y0, x1, y1):
    return [[x0, y0], [x1, y0], [x1, y1], [x0, y1]]


def make_cases():
    """case = dict(n_frames, lang, drop_score, deviation, area | None, tasks [[frame_no, cached?, default_area]], ocr {frame_no: [[quad, text, score]]})."""
    rng = np.random.default_rng(11)
    area = dict(ymin=24, ymax=38, xmin=4, xmax=60)
    inside = lambda: quad(8 + int(rng.integers(0, 4)), 26, 50, 36)
    C = []
    base_ocr = {
        1: [[inside(), "hello world", 0.93]],
        2: [[inside(), "hello world", 0.74]],                                   # below the confidence threshold
        3: [[quad(2, 26, 50, 36), "sticks out left", 0.99]],                    # overflow > 0
        4: [[quad(8, 2, 50, 12), "WATERMARK", 0.99], [inside(), "second line", 0.9]],
        5: [[quad(8, 2, 50, 12), "only outside", 0.99]],
        6: [],
        7: [[inside(), "中文字幕 with latin", 0.88], [quad(4, 24, 60, 38), "fills the area exactly", 0.8]],
        8: [[quad(60, 38, 70, 50), "touches the corner", 0.95]],                # intersection is a point: not empty, area 0
        9: [[inside(), "tab\tinside", 0.97]],
        10: [[inside(), "", 0.99], [inside(), "x", 0.751]],
    }
    seq = [[no, False, None] for no in range(1, 11)]
    for lang in ("ch", "en"):
        for dev in (0.0, 0.1):
            C.append(dict(n_frames=10, lang=lang, drop_score=0.75, deviation=dev, area=area, tasks=seq, ocr=base_ocr))
    C.append(dict(n_frames=10, lang="en", drop_score=0.75, deviation=0.0, area=None, tasks=seq, ocr=base_ocr))        # no area: keep all
    # half-frame crops of the fps sampler's default subtitle area (frame_preprocess) and cached detections of accurate mode
    lower = [[no, False, "LOWER_PART"] for no in (1, 4, 7)] + [[no, False, "UPPER_PART"] for no in (2, 5)] + [[3, False, "UNKNOWN"]]
    C.append(dict(n_frames=8, lang="ch", drop_score=0.75, deviation=0.0, area=None, tasks=lower, ocr=base_ocr))
    cached = [[1, True, None], [2, False, None], [4, True, None], [9, False, None], [12, False, None], [7, True, None]]   # 12: past the end
    C.append(dict(n_frames=10, lang="ch", drop_score=0.5, deviation=0.0, area=area, tasks=cached, ocr=base_ocr))
    # skewed quads whose box comes out "inverted" (xmin > xmax or ymin > ymax: max of one side's corners vs min of the other's,
    # ocr.py:118-129): the polygon's region does not depend on the corner order (found by tests/golden/fuzz_vs_reference.py)
    inv = {1: [[[[47.3, 26.0], [46.2, 26.2], [46.1, 34.0], [47.4, 34.3]], "x inverted", 0.9]],
           2: [[[[20.0, 33.8], [32.0, 33.6], [31.9, 32.8], [19.2, 33.6]], "y inverted", 0.9]],
           3: [[[[2.3, 26.0], [1.2, 26.2], [1.1, 34.0], [2.4, 34.3]], "inverted outside", 0.9]]}
    C.append(dict(n_frames=3, lang="ch", drop_score=0.75, deviation=0.0, area=area, tasks=[[no, False, None] for no in (1, 2, 3)], ocr=inv))
    # random boxes against the area filter (touching / containing / partial overlaps; integer and .5 coordinates)
    for k in range(6):
        ocr = {}
        for no in range(1, 13):
            items = []
            for _ in range(int(rng.integers(0, 4))):
                x0, y0 = int(rng.integers(0, 50)), int(rng.integers(14, 36))
                x1, y1 = x0 + int(rng.integers(2, 30)), y0 + int(rng.integers(2, 12))
                items.append([quad(x0, y0, x1, y1), "t%d_%d" % (no, len(items)), round(float(rng.uniform(0.6, 1.0)), 3)])
            ocr[no] = items
        C.append(dict(n_frames=12, lang="ch", drop_score=0.75, deviation=[0.0, 0.05, 0.3][k % 3], area=area,
                      tasks=[[no, False, None] for no in range(1, 13)], ocr=ocr))
    return C


def run_ocr_case(so, case):
    Area = sys.modules["backend.bean.subtitle_area"].SubtitleArea
    Const = sys.modules["backend.tools.constant"].SubtitleArea
    FakeCapture.frames = [frame_of(i + 1) for i in range(case["n_frames"])]
    seen = []

    def predict(self, img):
        no = int(img[0, 0, 0]) | (int(img[0, 0, 1]) << 8)
        seen.append([no, list(img.shape)])
        o = case["ocr"].get(no, [])
        return [q for q, _t, _s in o], [(t, s) for _q, t, s in o]
    sys.modules["backend.tools.ocr"].OcrRecogniser.predict = predict
    tq = queue.Queue()
    for no, use_cache, default_area in case["tasks"]:
        o = case["ocr"].get(no, [])
        dt, rr = ([q for q, _t, _s in o], [(t, s) for _q, t, s in o]) if use_cache else (None, None)
        tq.put((case["n_frames"], no, dt, rr, None, None if default_area is None else Const[default_area]))
    tq.put((case["n_frames"], -1, None, None, None, None))
    sub_area = None if case["area"] is None else Area(**case["area"])
    options = types.SimpleNamespace(REC_CHAR_TYPE=case["lang"], DROP_SCORE=case["drop_score"],
                                    SUB_AREA_DEVIATION_RATE=case["deviation"], DEBUG_OCR_LOSS=False, HARDWARD_ACCELERATOR=None)
    with tempfile.TemporaryDirectory() as td:
        raw = os.path.join(td, "raw.txt")
        so.subtitle_extract_handler(tq, queue.Queue(), os.path.join(td, "video.mp4"), raw, sub_area, options)
        text = open(raw, encoding="utf-8").read()
    return text, seen


def run_fps_cases():
    """extract_frame_by_fps through the reference's main.py (loaded with the stubs of make_frame_loop_golden)."""
    for k in [k for k in sys.modules if k == "backend" or k.startswith("backend.") or k in ("cv2", "tools", "tools.reformat")]:
        del sys.modules[k]
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import make_frame_loop_golden as G
    main = G.install_stubs(80)
    out = []
    for n_frames, fps, freq in [(30, 30.0, 3), (31, 25.0, 3), (100, 23.976, 3), (10, 60.0, 1), (7, 24.0, 24), (9, 24.0, 30),
                                (50, 29.97, 2), (0, 25.0, 3), (1, 25.0, 3), (64, 12.0, 5)]:
        ext = object.__new__(main.SubtitleExtractor)
        ext.frame_count = n_frames
        ext.fps = fps
        tasks = []

        class Cap:
            def __init__(self):
                self.i, self.open = 0, True

            def isOpened(self):
                return self.open

            def read(self):
                if self.i >= n_frames:
                    return False, None
                self.i += 1
                return True, self.i

            def release(self):
                self.open = False

        class Q:
            def put(self, task):
                tasks.append([task[0], task[1], task[5]])
        main.config.extractFrequency = _Val(freq)
        main.config.subtitleArea = _Val("AREA")
        ext.video_cap = Cap()
        ext.subtitle_ocr_task_queue = Q()
        ext.update_progress = lambda **k: None
        ext.extract_frame_by_fps()
        out.append(dict(n_frames=n_frames, fps=fps, extract_frequency=freq, tasks=tasks))
    return out


def main():
    install_common()
    so = load("backend.tools.subtitle_ocr", "backend/tools/subtitle_ocr.py")
    out = {"source": "backend/tools/subtitle_ocr.py:20-85,126-289 and backend/main.py:228-253 @ v2.2.0 executed with stubbed "
                     "third-party imports", "frame_shape": [H, W, 3], "ocr_cases": [], "fps_cases": []}
    for case in make_cases():
        raw, seen = run_ocr_case(so, case)
        c = dict(case)
        c["ocr"] = {str(k): v for k, v in case["ocr"].items()}
        c["raw"] = raw
        c["seen"] = seen
        out["ocr_cases"].append(c)
    out["fps_cases"] = run_fps_cases()
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(out, f, separators=(",", ":"), ensure_ascii=False)
    print("wrote", OUT, len(out["ocr_cases"]), "ocr cases,", sum(len(c["raw"].splitlines()) for c in out["ocr_cases"]), "raw lines;",
          len(out["fps_cases"]), "fps cases")


if __name__ == "__main__":
    main()
