#!/usr/bin/env python3
"""Generates tests/golden/frame_loop.json by RUNNING the reference's own accurate-mode frame loop
(SubtitleExtractor.extract_frame_by_det + _compare_ocr_result + __get_area_text, backend/main.py:255-376,906-952)
in this container on scripted detector / recogniser outputs.

backend/main.py imports cv2, Levenshtein, pysrt, paddle, qfluentwidgets, ... none installed: they are replaced by
inert stubs.  Two stubs carry behaviour and are therefore part of what the vectors pin:
  * Levenshtein.ratio -> the normalised InDel similarity 2*LCS(a,b)/(len(a)+len(b)) (1.0 for two empty strings),
    i.e. the published definition of Levenshtein==0.26.0's ratio (requirements.txt:2);
  * the detector / recogniser / video capture are scripted fakes (the networks are not what is under test here).
The state machine, cache eviction, queue-flush quirks and get_coordinates are the reference's code.
Only inputs and outputs are written (data, not source).  Needs /root/reference; not run on the GPU box.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frame_loop.json")


def lcs_ratio(a, b):
    if not a and not b:
        return 1.0
    prev = [0] * (len(b) + 1)
    for ca in a:
        cur = [0]
        for j, cb in enumerate(b):
            cur.append(prev[j] + 1 if ca == cb else max(prev[j + 1], cur[j]))
        prev = cur
    return 2.0 * prev[-1] / (len(a) + len(b))


class _Val:
    def __init__(self, v):
        self.value = v


def install_stubs(threshold):
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Tqdm:
        def __init__(self, *a, **k):
            pass

        def update(self, n):
            pass

        @staticmethod
        def write(*a, **k):
            pass

    mod("cv2", VideoCapture=object)
    mod("Levenshtein", ratio=lcs_ratio)
    mod("pysrt")
    mod("paddle")
    mod("paddleocr", PaddleOCR=object)
    mod("tqdm", tqdm=_Tqdm)
    mod("shapely")
    mod("shapely.geometry", Polygon=object)
    cfg = types.SimpleNamespace(thresholdTextSimilarity=_Val(threshold), subtitleArea=_Val("AREA"),
                                hardwareAcceleration=_Val(True))
    mod("backend.config", config=cfg, tr={}, BASE_DIR=REF, __all__=["config", "tr", "BASE_DIR"])

    class _HA:
        onnx_providers = []

        @classmethod
        def instance(cls):
            return cls()

        def has_cuda(self):
            return False

        def set_enabled(self, e):
            pass

    mod("backend.tools.hardware_accelerator", HardwareAccelerator=_HA)
    mod("backend.tools.paddle_model_config", PaddleModelConfig=object)
    mod("backend.tools.process_manager", ProcessManager=object)
    mod("backend.tools.subtitle_detect", SubtitleDetect=object)
    mod("backend.tools.subtitle_ocr")
    mod("tools")
    mod("tools.reformat")
    pkg = mod("backend")
    pkg.__path__ = [os.path.join(REF, "backend")]
    tools = mod("backend.tools")
    tools.__path__ = [os.path.join(REF, "backend", "tools")]
    tools.subtitle_ocr = sys.modules["backend.tools.subtitle_ocr"]
    sys.modules["tools"].reformat = sys.modules["tools.reformat"]
    bean = mod("backend.bean")
    bean.__path__ = [os.path.join(REF, "backend", "bean")]

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m
    load("backend.tools.ocr", "backend/tools/ocr.py")                    # the reference's get_coordinates
    load("backend.bean.subtitle_area", "backend/bean/subtitle_area.py")
    return load("backend.main_ref", "backend/main.py")


AREA = dict(ymin=800, ymax=1000, xmin=100, xmax=1800)


def quad(x0, y0, x1, y1):
    return [[x0, y0], [x1, y0], [x1, y1], [x0, y1]]


def make_scenarios():
    """A scenario = list of per-frame dicts {"boxes": [[4x2]...], "ocr": [[quad, text, score]...]}."""
    rng = np.random.default_rng(7)
    inside = lambda: quad(300 + int(rng.integers(0, 50)), 850, 1500, 920)
    outside = lambda: quad(300, 100, 900, 160)
    straddle = lambda: quad(50, 850, 900, 920)            # sticks out of the area on the left

    def frame(text=None, extra_outside=False, only_outside=False, straddling=False, score=0.95, ocr_extra=None):
        boxes, ocr = [], []
        if text is not None and not only_outside:
            q = straddle() if straddling else inside()
            boxes.append(q)
            ocr.append([q, text, score])
        if extra_outside or only_outside:
            q = outside()
            boxes.append(q)
            ocr.append([q, "WATERMARK", 0.99])
        if ocr_extra:
            ocr += ocr_extra
        return {"boxes": boxes, "ocr": ocr}

    S = []
    # 1: one subtitle in the middle
    S.append([frame()] * 3 + [frame("hello world")] * 5 + [frame()] * 3)
    # 2: two subtitles back to back with different text (end detected by ratio <= threshold)
    S.append([frame()] * 2 + [frame("first line of text")] * 4 + [frame("something else entirely")] * 4 + [frame()] * 2)
    # 3: near-identical OCR noise (ratio > 0.8 keeps the subtitle), then a gap, then the same text again
    S.append([frame("the quick brown fox")] * 3 + [frame("the quick brown f0x")] * 2 + [frame()] + [frame("the quick brown fox")] * 3)
    # 4: subtitle runs until the very last frame
    S.append([frame()] * 2 + [frame("until the end")] * 6)
    # 5: only out-of-area boxes / straddling boxes never start a subtitle; mixed in-area + watermark does
    S.append([frame(only_outside=True)] * 3 + [frame("x", straddling=True)] * 2 + [frame("real one", extra_outside=True)] * 4 + [frame(only_outside=True)] * 2)
    # 6: long subtitle (cache eviction of entries older than 10 frames) followed by a change
    S.append([frame("a long lasting subtitle line")] * 30 + [frame("next")] * 3 + [frame()] * 2)
    # 7: single-frame subtitles separated by single empty frames
    S.append([frame("a"), frame(), frame("b"), frame(), frame("c"), frame()])
    # 8: empty video / no text at all
    S.append([frame()] * 5)
    # 9: text that shrinks gradually (ratio vs the START frame, not the previous one)
    base = "abcdefghijklmnopqrstuvwxyz"
    S.append([frame(base[:26 - k]) for k in range(12)] + [frame()] * 2)
    # 10+: random timelines
    words = ["alpha", "beta gamma", "delta", "epsilon zeta eta", "theta", "iota kappa"]
    for _ in range(12):
        tl = []
        while len(tl) < 40:
            r = rng.random()
            n = int(rng.integers(1, 7))
            if r < 0.35:
                tl += [frame()] * n
            elif r < 0.45:
                tl += [frame(only_outside=True)] * n
            else:
                t = str(rng.choice(words))
                if rng.random() < 0.3:
                    t2 = t[:-1] + "#"
                    tl += [frame(t)] * max(1, n // 2) + [frame(t2)] * max(1, n - n // 2)
                else:
                    tl += [frame(t, extra_outside=rng.random() < 0.3)] * n
        S.append(tl[:40])
    return S


def run_reference(main, scenario, threshold):
    ext = object.__new__(main.SubtitleExtractor)
    ext.sub_area = types.SimpleNamespace(**AREA)
    ext.frame_count = len(scenario)
    ext.ocr = None
    tasks = []

    class Cap:
        def __init__(self):
            self.i = 0
            self.open = True

        def isOpened(self):
            return self.open

        def read(self):
            if self.i >= len(scenario):
                return False, None
            self.i += 1
            return True, self.i            # the "frame" is its 1-based number

        def release(self):
            self.open = False

    class Det:
        def detect_subtitle(self, frame_no):
            b = scenario[frame_no - 1]["boxes"]
            return (np.asarray(b, dtype=np.float32).reshape(-1, 4, 2), 0.0)

    class Ocr:
        calls = []

        def predict(self, frame_no):
            Ocr.calls.append(frame_no)
            o = scenario[frame_no - 1]["ocr"]
            return [q for q, _t, _s in o], [(t, s) for _q, t, s in o]

    class Q:
        def put(self, task):
            total, no, dt_box, rec_res, ms, area = task
            tasks.append({"total": total, "frame_no": no, "cached": dt_box is not None,
                          "texts": None if rec_res is None else [t for t, _s in rec_res]})
    ext.video_cap = Cap()
    ext.sub_detector = Det()
    ext.ocr = Ocr()
    Ocr.calls = []
    ext.subtitle_ocr_task_queue = Q()
    ext.update_progress = lambda **k: None
    ext.extract_frame_by_det()
    return tasks, list(Ocr.calls)


def main():
    threshold = 80
    m = install_stubs(threshold)
    out = {"source": "backend/main.py:255-376,906-952 @ v2.2.0 executed with stubbed third-party imports",
           "area": AREA, "threshold": threshold, "scenarios": []}
    for sc in make_scenarios():
        tasks, calls = run_reference(m, sc, threshold)
        out["scenarios"].append({"frames": sc, "tasks": tasks, "predict_calls": calls})
    out["ratio_cases"] = [[a, b, lcs_ratio(a, b)] for a, b in
                          [("", ""), ("a", ""), ("abc", "abc"), ("the quick brown fox", "the quick brown f0x"),
                           ("first line of text", "something else entirely"), ("kitten", "sitting")]]
    with open(OUT, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", OUT, len(out["scenarios"]), "scenarios;", sum(len(s["tasks"]) for s in out["scenarios"]), "tasks")


if __name__ == "__main__":
    main()
