#!/usr/bin/env python3
"""Generates tests/golden/raw_filters.json by RUNNING the reference's own raw.txt filters
(SubtitleExtractor._unite_coordinates, _detect_watermark_area, _detect_subtitle_area, filter_watermark,
filter_scene_text; backend/main.py:506-612,671-729,866-881,954-963) in this container on scripted raw files.

Same stub set as make_frame_loop_golden.py.  Stubs that carry behaviour and are therefore part of what the vectors pin:
  * builtins.input -> scripted answers ('y' / 'n') per question, in the order the reference asks;
  * cv2.VideoCapture -> always delivers a black 1080p frame; blur / rectangle / putText / imwrite do nothing (the
    reference only draws the candidate areas for the user to look at);
  * config.tolerantPixelX / tolerantPixelY -> plain ints (the reference compares them without `.value`).
Only inputs and outputs are written (data, not source).  Needs /root/reference; not run on the GPU box.
"""
import builtins
import collections
import json
import os
import random
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_frame_loop_golden as G  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raw_filters.json")


def raw(frame_no, text, coord):
    return f"{str(frame_no).zfill(8)}\t{coord}\t{text}\n"


def make_scenarios():
    rnd = random.Random(7)
    S = []
    # 1: a fixed logo (jittering by a few px) on every frame + subtitles whose box moves with the text length
    lines = []
    for n in range(1, 60):
        lines.append(raw(n, "TV5", (1700 + rnd.randint(-3, 3), 1850 + rnd.randint(-3, 3), 40 + rnd.randint(-2, 2), 90 + rnd.randint(-2, 2))))
        if n % 3:
            w = rnd.choice([200, 420, 650])
            lines.append(raw(n, f"subtitle {n // 10}", (960 - w, 960 + w, 900 + rnd.randint(-4, 4), 960 + rnd.randint(-4, 4))))
    S.append(dict(lines=lines, answers=["y", "n", "n", "n", "n", "y"]))
    # 2: same file, user keeps everything
    S.append(dict(lines=lines, answers=["n"] * 5 + ["n"]))
    # 3: more than five distinct areas, chains of similarity (a~b, b~c, a!~c), ties in the counts
    lines = []
    k = 0
    for base in [(100, 300, 100, 140), (190, 390, 100, 140), (280, 480, 100, 140), (900, 1300, 500, 560),
                 (900, 1300, 545, 605), (50, 90, 1000, 1040), (1500, 1800, 20, 60), (600, 800, 700, 760)]:
        for rep in range(3 + (k % 2)):
            k += 1
            lines.append(raw(k, f"t{k}", (base[0] + rep, base[1] - rep, base[2] + rep, base[3])))
    rnd.shuffle(lines)
    S.append(dict(lines=lines, answers=["y", "y", "n", "y", "n", "y"]))
    # 4: scene text above and below the band; band widened by 50 px; abs() of a negative ymin
    lines = [raw(n, "sub", (400, 1500, 30, 80)) for n in range(1, 20)] + [raw(5, "sign", (100, 300, 0, 25)),
             raw(6, "far", (100, 300, 131, 170)), raw(7, "edge", (100, 300, 20, 130)), raw(8, "edge2", (100, 300, 19, 130)),
             raw(9, "edge3", (100, 300, 20, 131))]
    S.append(dict(lines=lines, answers=["n"] * 5 + ["y"]))
    # 5: a text with a tab and a text that contains another area's repr
    lines = [raw(1, "a\tb", (10, 20, 30, 40)), raw(2, "plain", (10, 20, 30, 40)), raw(3, "mentions (10, 20, 30, 40) in text", (500, 900, 600, 650)),
             raw(4, "other", (500, 900, 600, 650)), raw(5, "other", (500, 900, 600, 650))]
    S.append(dict(lines=lines, answers=["n", "y", "y"]))
    # 6: one line only
    S.append(dict(lines=[raw(1, "only", (1, 2, 3, 4))], answers=["y", "y"]))
    # 7: large random file (exercises the in-place walk of _unite_coordinates)
    lines = []
    for n in range(400):
        x = rnd.choice([100, 180, 260, 340, 800, 1500]) + rnd.randint(-30, 30)
        y = rnd.choice([100, 140, 180, 900, 940]) + rnd.randint(-15, 15)
        lines.append(raw(n, f"w{n % 7}", (x, x + rnd.choice([200, 290, 400]), y, y + rnd.choice([40, 80]))))
    S.append(dict(lines=lines, answers=["y", "n", "y", "n", "y", "y"]))
    # 8: a text with a tab in the MIDDLE of the file: the rewrite drops its tail and its newline, so the row shares a line
    # with the next one, and the merged line goes when EITHER of its two areas is deleted (found by tests/golden/fuzz_vs_reference.py)
    lines = [raw(1, "logo", (1700, 1850, 40, 90)), raw(2, "keep\tlost", (300, 900, 600, 650)), raw(3, "logo", (1700, 1850, 40, 90)),
             raw(4, "sub", (300, 900, 600, 650)), raw(5, "x\ty", (100, 200, 900, 950)), raw(6, "sub", (300, 900, 600, 650)),
             raw(7, "logo", (1700, 1850, 40, 90)), raw(8, "sub", (300, 900, 600, 650))]
    S.append(dict(lines=lines, answers=["n", "y", "n", "y"]))
    S.append(dict(lines=lines, answers=["y", "n", "y", "y"]))
    return S


def run_reference(main, sc):
    class Cap:
        def __init__(self, path):
            pass

        def set(self, prop, v):
            pass

        def read(self):
            return True, np.zeros((1080, 1920, 3), np.uint8)

        def release(self):
            pass
    main.cv2.VideoCapture = Cap
    main.cv2.CAP_PROP_POS_FRAMES = 1
    main.cv2.FONT_HERSHEY_SIMPLEX = 0
    main.cv2.LINE_AA = 16
    main.cv2.blur = lambda img, k: img
    main.cv2.rectangle = lambda *a, **k: None
    main.cv2.putText = lambda *a, **k: None
    main.cv2.imwrite = lambda *a, **k: True
    main.config.tolerantPixelX = 100
    main.config.tolerantPixelY = 50
    main.config.subtitleAreaDeviationPixel = G._Val(50)
    main.config.waterarkAreaNum = G._Val(5)
    res = {}
    asked = []
    answers = list(sc["answers"])

    def fake_input(prompt=""):
        asked.append(prompt)
        return answers.pop(0)
    real_input = builtins.input
    builtins.input = fake_input
    try:
        with tempfile.TemporaryDirectory() as d:
            ext = object.__new__(main.SubtitleExtractor)
            ext.raw_subtitle_path = os.path.join(d, "raw.txt")
            ext.frame_output_dir = os.path.join(d, "frames")
            ext.video_path = "video.mp4"
            ext.frame_count = 1000
            ext.append_output = lambda *a, **k: None

            def put():
                with open(ext.raw_subtitle_path, "w", encoding="utf-8") as f:
                    f.writelines(sc["lines"])

            def get():
                with open(ext.raw_subtitle_path, encoding="utf-8") as f:
                    return f.read()
            put()
            areas = ext._detect_watermark_area()
            res["watermark_areas"] = [[list(a), c] for a, c in areas]
            res["raw_after_detect"] = get()
            res["subtitle_area"] = [[list(a), c] for a, c in ext._detect_subtitle_area()]
            put()
            ext.filter_watermark()
            res["n_watermark_questions"] = len(asked)
            res["raw_after_watermark"] = get()
            try:
                ext.filter_scene_text()
                res["scene_text_error"] = None
            except Exception as e:                      # an emptied file makes the reference raise: pinned as such
                res["scene_text_error"] = type(e).__name__
            res["raw_after_scene_text"] = get()
    finally:
        builtins.input = real_input
    return res


def main():
    m = G.install_stubs(80)
    m.tr = collections.defaultdict(lambda: collections.defaultdict(lambda: "{}"))
    out = {"source": "backend/main.py:506-612,671-729,866-881,954-963 @ v2.2.0 executed with stubbed third-party imports",
           "scenarios": []}
    for sc in make_scenarios():
        rec = dict(lines=sc["lines"], answers=sc["answers"])
        rec.update(run_reference(m, sc))
        out["scenarios"].append(rec)
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, separators=(",", ":"))
    print("wrote", OUT, len(out["scenarios"]), "scenarios")


if __name__ == "__main__":
    main()
