#!/usr/bin/env python3
"""Differential fuzzing of the host-side text logic against the REFERENCE'S OWN code, executed in this container with the
stub set of the golden generators (make_srt_golden / make_raw_filters_golden / make_text_cleanup_golden).  Random scenarios
go through backend/main.py (raw.txt -> SRT; watermark / scene-text filters) and backend/tools/reformat.py, and through
vse_amd.srt / raw_filters / text_cleanup; any difference is printed with the scenario that caused it.

Needs /root/reference (build container only; nothing here runs on the GPU box).  Scenarios that expose a difference are added
to the golden generators so that the committed vectors pin the fix.
usage: python tests/golden/fuzz_vs_reference.py [--cases 300] [--seed 0] [--only srt|filters|cleanup]"""
import argparse
import collections
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

VOCAB = ["hello world", "the quick brown fox", "subtitle", "你好，世界", "再见", "ｆｕｌｌ　ｗｉｄｔｈ", "a", "", " ", "I'm here", "l'm here",
         "x" * 40, "Dr. Who", "① ②", "ﬁne", "line one", "abcde", "abcdX", "abcXY", "12:30", "A-B", "50 %", "end."]


def mutate(rnd, t):
    if not t or rnd.random() < 0.5:
        return t
    k = rnd.randrange(len(t))
    op = rnd.choice("idr s")
    if op == "i":
        return t[:k] + rnd.choice("abcxyz0 ") + t[k:]
    if op == "d":
        return t[:k] + t[k + 1:]
    if op == "r":
        return t[:k] + rnd.choice("abcxyz0") + t[k + 1:]
    if op == "s":
        return t.replace(" ", "")
    return t + " "


def rand_raw(rnd, vocab=VOCAB):
    lines = []
    no = rnd.randrange(0, 20)
    for _ in range(rnd.randrange(0, 12)):                      # subtitles
        text = rnd.choice(vocab)
        for _ in range(rnd.choice([1, 1, 2, 5, 30, 60])):      # frames it stays
            for _ in range(rnd.choice([1, 1, 1, 2, 3])):       # lines on this frame
                x0, y0 = rnd.randrange(0, 1500), rnd.randrange(0, 1000)
                coord = (x0, x0 + rnd.randrange(1, 400), y0, y0 + rnd.randrange(1, 80))
                lines.append(f"{str(no).zfill(8)}\t{coord}\t{mutate(rnd, text)}\n")
            no += rnd.choice([1, 1, 1, 2, 7])
        no += rnd.choice([0, 1, 10, 100])
    return lines


def fuzz_srt(n, seed):
    import make_frame_loop_golden as G
    import make_srt_golden as S
    from vse_amd import srt
    m = G.install_stubs(80)
    m.tr = collections.defaultdict(lambda: collections.defaultdict(lambda: "{}"))
    rnd = random.Random(seed)
    bad = 0
    for i in range(n):
        sc = dict(fps=rnd.choice([23.976, 24.0, 25.0, 29.97, 30.0, 60.0, 12.5]), lines=rand_raw(rnd), threshold=rnd.choice([80, 80, 60, 95, 100, 0]))
        if rnd.random() < 0.4:
            top = max([int(ln[:8]) for ln in sc["lines"]] + [0]) + 70
            step = 1000.0 / sc["fps"]
            sc["msec"] = {k: round(k * step * rnd.choice([1.0, 1.0, 1.001])) for k in range(0, top) if rnd.random() < 0.97}
        try:
            ref = S.run_reference(m, sc)
        except Exception as e:                                      # noqa: BLE001 - the reference itself fails: compare behaviour
            ref = ("EXC", type(e).__name__)
        tbl = sc.get("msec")
        pos = (lambda no: float(tbl[no]) if no in tbl else None) if tbl is not None else (lambda no: None)
        try:
            text, short, norm = srt.generate_subtitle_file(sc["lines"], sc["fps"], sc["threshold"], pos)
            mine = (text, short, "".join(norm))
        except Exception as e:                                      # noqa: BLE001
            mine = ("EXC", type(e).__name__)
        if mine != ref:
            bad += 1
            if bad <= 5:
                print("SRT DIFF", {k: v for k, v in sc.items() if k != "msec"}, "msec" in sc)
                print("  ref :", repr(ref)[:600])
                print("  mine:", repr(mine)[:600])
    print(f"srt: {n} scenarios, {bad} differences")
    return bad


def fuzz_filters(n, seed):
    import make_frame_loop_golden as G
    import make_raw_filters_golden as F
    from vse_amd import raw_filters
    m = G.install_stubs(80)
    m.tr = collections.defaultdict(lambda: collections.defaultdict(lambda: "{}"))
    rnd = random.Random(seed)
    bad = 0
    for i in range(n):
        lines = []
        anchors = [(rnd.randrange(0, 1600), rnd.randrange(0, 1000)) for _ in range(rnd.randrange(1, 9))]
        for k in range(rnd.randrange(1, 120)):
            ax, ay = rnd.choice(anchors)
            x, y = ax + rnd.randint(-40, 40), ay + rnd.randint(-25, 25)
            coord = (x, x + rnd.choice([60, 200, 290, 400]) + rnd.randint(-5, 5), y, y + rnd.choice([30, 40, 80]) + rnd.randint(-3, 3))
            text = rnd.choice(["w", "logo", "sub title", "a\tb", "(1, 2, 3, 4)", "", "字幕"]) + str(k % 5)
            lines.append(f"{str(rnd.randrange(1, 400) if rnd.random() < 0.2 else k + 1).zfill(8)}\t{coord}\t{text}\n")
        sc = dict(lines=lines, answers=[rnd.choice("yn") for _ in range(8)])
        try:
            ref = F.run_reference(m, sc)
        except Exception as e:                                      # noqa: BLE001
            ref = {"exc": type(e).__name__}
        try:
            mine = {}
            areas, out = raw_filters.detect_watermark_area(sc["lines"])
            mine["watermark_areas"] = [[list(x), c] for x, c in areas]
            mine["raw_after_detect"] = "".join(out)
            mine["subtitle_area"] = [[list(x), c] for x, c in raw_filters.detect_subtitle_area(mine["raw_after_detect"].splitlines(keepends=True))]
            answers = list(sc["answers"])
            asked = []

            def decide(x):
                asked.append(x)
                return answers.pop(0) == "y"
            with tempfile.TemporaryDirectory() as d:
                rawp = os.path.join(d, "raw.txt")
                with open(rawp, "w", encoding="utf-8") as f:
                    f.writelines(sc["lines"])
                raw_filters.filter_watermark_file(rawp, decide)
                mine["n_watermark_questions"] = len(asked)
                mine["raw_after_watermark"] = open(rawp, encoding="utf-8").read()
                try:
                    raw_filters.filter_scene_text_file(rawp, decide)
                    mine["scene_text_error"] = None
                except Exception as e:                              # noqa: BLE001
                    mine["scene_text_error"] = type(e).__name__
                mine["raw_after_scene_text"] = open(rawp, encoding="utf-8").read()
        except Exception as e:                                      # noqa: BLE001
            mine = {"exc": type(e).__name__ + ": " + str(e)[:80]}
        keys = set(ref) | set(mine)
        diff = [k for k in keys if ref.get(k) != mine.get(k)]
        if diff:
            bad += 1
            if bad <= 5:
                print("FILTER DIFF in", diff, "lines:", len(lines), "answers", sc["answers"])
                for k in diff[:3]:
                    print("  ref ", k, repr(ref.get(k))[:300])
                    print("  mine", k, repr(mine.get(k))[:300])
    print(f"filters: {n} scenarios, {bad} differences")
    return bad


def fuzz_cleanup(n, seed):
    import json
    import make_text_cleanup_golden as T
    from vse_amd import text_cleanup
    m = T.install()
    typo = json.load(open(os.path.join(T.REF, "backend", "configs", "typoMap.json"), encoding="utf-8"))
    rnd = random.Random(seed)
    words = ["hello", "world", "i", "im", "I'm", "l'm", "dont", "don't", "go", "going", "Let'sqo", "Iife", "is", "it", "its", "it's",
             "what", "whats", "you", "youre", "10", "%", "-made", "self", "Dr.", "Smith", "NEW", "York", "你好", "世界", "威筋", "。", "，",
             ",", ".", "!", "?", "·", "\"", "'", "  ", " ", "\n", "\n ", "\ni", "x", "abc", "2021", "men", "and", "to", "be"]
    bad = 0
    for i in range(n):
        texts = []
        for _ in range(rnd.randrange(1, 8)):
            t = "".join(rnd.choice(words) + rnd.choice(["", "", " ", " "]) for _ in range(rnd.randrange(0, 9)))
            texts.append(t.replace("\n\n", "\n").strip("\n"))
        lang = rnd.choice(["en", "ch", "ch_tra", "japan"])
        src = T.make_srt(texts)
        with tempfile.TemporaryDirectory() as td:
            p1, p2 = os.path.join(td, "a.srt"), os.path.join(td, "b.srt")
            open(p1, "w", encoding="utf-8").write(src)
            open(p2, "w", encoding="utf-8").write(src)
            try:
                ok = m.execute(p1, lang)
                ref = open(p1, encoding="utf-8").read()
            except Exception as e:                                  # noqa: BLE001
                ref = "EXC " + type(e).__name__
            try:
                text_cleanup.execute(p2, lang, segment=T.scripted_segment, typo_map=typo)
                mine = open(p2, encoding="utf-8").read()
            except Exception as e:                                  # noqa: BLE001
                mine = "EXC " + type(e).__name__ + str(e)[:60]
        if ref != mine:
            bad += 1
            if bad <= 6:
                print("CLEANUP DIFF lang", lang, "texts", texts)
                print("  ref :", repr(ref)[:500])
                print("  mine:", repr(mine)[:500])
    print(f"cleanup: {n} files, {bad} differences")
    return bad


def fuzz_frame_loop(n, seed):
    import numpy as np
    from types import SimpleNamespace
    import make_frame_loop_golden as G
    from oracle import frame_loop_ref as R
    from vse_amd import frame_select
    m = G.install_stubs(80)
    m.tr = collections.defaultdict(lambda: collections.defaultdict(lambda: "{}"))
    rnd = random.Random(seed)
    A = G.AREA
    bad = 0

    def rquad():
        kind = rnd.choice(["in", "in", "in", "out", "straddle", "edge", "tiny"])
        if kind == "in":
            x0, y0 = rnd.randint(A["xmin"], A["xmin"] + 200), rnd.randint(A["ymin"], A["ymin"] + 20)
            return G.quad(x0, y0, rnd.randint(x0 + 10, A["xmax"]), rnd.randint(y0 + 5, A["ymax"]))
        if kind == "out":
            return G.quad(300, 100, 900, 160)
        if kind == "straddle":
            return G.quad(A["xmin"] - rnd.randint(1, 60), A["ymin"], A["xmin"] + 300, A["ymax"])
        if kind == "edge":
            return G.quad(A["xmin"], A["ymin"], A["xmax"], A["ymax"])
        return G.quad(A["xmin"] + 5, A["ymin"] + 5, A["xmin"] + 6, A["ymin"] + 6)
    words = ["alpha", "beta gamma", "delta", "epsilon zeta eta", "theta", "", " ", "alpha!", "alphb", "ALPHA", "你好", "beta  gamma"]

    def norm(tasks):
        return [{"total": t[0], "frame_no": t[1], "cached": t[2] is not None,
                 "texts": None if t[3] is None else [x[0] for x in t[3]]} for t in tasks]
    for i in range(n):
        tl = []
        L = rnd.choice([0, 1, 2, 5, 12, 40, 90])
        while len(tl) < L:
            k = rnd.randint(1, 14)
            boxes = [rquad() for _ in range(rnd.choice([0, 0, 1, 1, 1, 2, 3]))]
            t = rnd.choice(words)
            for _ in range(k):
                ocr = []
                for q in boxes:
                    if rnd.random() < 0.9:                  # the recogniser may find nothing where the detector saw a box
                        ocr.append([q, t if rnd.random() < 0.8 else rnd.choice(words), rnd.choice([0.95, 0.5, 0.76, 0.74])])
                if rnd.random() < 0.1:
                    ocr.append([rquad(), "extra", 0.9])     # ... or an extra line
                tl.append({"boxes": boxes if rnd.random() < 0.9 else [], "ocr": ocr})
        tl = tl[:L]
        thr = rnd.choice([80, 80, 60, 100, 0])
        m.config.thresholdTextSimilarity = G._Val(thr)
        try:
            ref_tasks, ref_calls = G.run_reference(m, tl, thr)
        except Exception as e:                                      # noqa: BLE001
            ref_tasks, ref_calls = "EXC " + type(e).__name__, None

        def fakes():
            calls = []

            def detect(no):
                return np.asarray(tl[no - 1]["boxes"], dtype=np.float32).reshape(-1, 4, 2)

            def predict(no):
                calls.append(no)
                o = tl[no - 1]["ocr"]
                return [q for q, _t, _s in o], [(t_, s_) for _q, t_, s_ in o]
            return detect, predict, calls
        outs = {}
        try:
            detect, predict, calls = fakes()
            outs["oracle"] = (norm(R.extract_frame_by_det(range(1, L + 1), L, detect, predict, A, thr)), calls)
        except Exception as e:                                      # noqa: BLE001
            outs["oracle"] = ("EXC " + type(e).__name__, None)
        for chunk, pre in ((1, False), (7, False), (64, True), (5, True)):
            try:
                detect, predict, calls = fakes()
                sel = frame_select.AccurateFrameSelector(lambda fs: [detect(f) for f in fs], predict, SimpleNamespace(**A), L, thr,
                                                         chunk=chunk, predict_batch=(lambda fs: [predict(f) for f in fs]) if pre else None)
                outs[f"selector chunk={chunk} prefetch={pre}"] = (norm(sel.run(range(1, L + 1))), None if pre else calls)
            except Exception as e:                                  # noqa: BLE001
                outs[f"selector chunk={chunk} prefetch={pre}"] = ("EXC " + type(e).__name__ + str(e)[:60], None)
        for name, (tasks, calls) in outs.items():
            if tasks != ref_tasks or (calls is not None and ref_calls is not None and calls != ref_calls):
                bad += 1
                if bad <= 5:
                    print("FRAME LOOP DIFF", name, "len", L, "thr", thr)
                    print("  timeline:", [(len(f["boxes"]), [o[1] for o in f["ocr"]]) for f in tl][:60])
                    print("  ref  tasks:", str(ref_tasks)[:400], "calls", ref_calls)
                    print("  mine tasks:", str(tasks)[:400], "calls", calls)
                break
    print(f"frame loop: {n} timelines, {bad} differences")
    return bad


def fuzz_glue(n, seed):
    """backend/tools/ocr.py: OcrRecogniser.predict's box conversion / line grouping / ordering and get_coordinates."""
    import importlib.util
    import numpy as np
    import make_ocr_glue_golden as O
    from oracle import pipeline_ref as P
    from vse_amd import shim
    O.install_stubs()
    spec = importlib.util.spec_from_file_location("backend.tools.ocr", os.path.join(O.REF, "backend/tools/ocr.py"))
    ocr = importlib.util.module_from_spec(spec)
    sys.modules["backend.tools.ocr"] = ocr
    spec.loader.exec_module(ocr)
    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(n):
        k = int(rng.integers(0, 10))
        boxes, texts = [], []
        for j in range(k):
            kind = rng.choice(["rect", "quad", "int", "tie", "neg"])
            x0, y0 = float(rng.uniform(0, 1800)), float(rng.choice([rng.uniform(0, 1000), 905.0, 895.0, 900.4999, 15.0, 5.0]))
            w, h = float(rng.uniform(1, 600)), float(rng.uniform(1, 90))
            q = np.array([[x0, y0], [x0 + w, y0], [x0 + w, y0 + h], [x0, y0 + h]], np.float32)
            if kind == "quad":
                q += rng.uniform(-8, 8, (4, 2)).astype(np.float32)
            elif kind == "int":
                q = np.rint(q)
            elif kind == "tie":
                q[:, 1] = np.rint(q[:, 1] / 5) * 5
            elif kind == "neg":
                q -= np.float32(rng.uniform(0, 40))
            boxes.append(q)
            texts.append((f"t{j}", float(rng.uniform(0, 1))))
        rec = ocr.OcrRecogniser()
        rec.recogniser = lambda image, cls=False: (boxes, texts, None)
        try:
            dt, res = rec.predict(None)
            ref = ([[tuple(map(int, p)) for p in b] for b in dt] if k else (dt, res), [(t, float(s_)) for t, s_ in res] if k else None,
                   [tuple(map(int, c)) for c in ocr.get_coordinates(boxes)])
        except Exception as e:                                      # noqa: BLE001
            ref = "EXC " + type(e).__name__
        for name, arrange, coords in (("shim", shim.OcrRecogniser.arrange, shim.get_coordinates), ("oracle", P.ocr_predict_glue, P.get_coordinates)):
            try:
                dt2, res2 = arrange(boxes, texts)
                mine = ([[tuple(map(int, p)) for p in b] for b in dt2] if k else (dt2, res2), [(t, float(s_)) for t, s_ in res2] if k else None,
                        [tuple(map(int, c)) for c in coords(boxes)])
            except Exception as e:                                  # noqa: BLE001
                mine = "EXC " + type(e).__name__ + str(e)[:60]
            if mine != ref:
                bad += 1
                if bad <= 5:
                    print("GLUE DIFF", name, [b.tolist() for b in boxes])
                    print("  ref :", str(ref)[:500])
                    print("  mine:", str(mine)[:500])
                break
    print(f"glue: {n} cases, {bad} differences")
    return bad


def fuzz_extract(n, seed):
    """backend/tools/subtitle_ocr.py: task producer / consumer + extract_subtitles (area / confidence / language filters)."""
    import numpy as np
    import make_extract_golden as E
    from vse_amd import extractor
    E.install_common()
    so = E.load("backend.tools.subtitle_ocr", "backend/tools/subtitle_ocr.py")
    rng = np.random.default_rng(seed)
    H, W = E.H, E.W
    bad = 0

    def frame_of(no):
        f = np.zeros((H, W, 3), np.uint8)
        f[:, :, 0] = no & 255
        f[:, :, 1] = no >> 8
        return f

    class Ocr:
        def __init__(self, script):
            self.script = script

        def predict(self, img):
            no = int(img[0, 0, 0]) | (int(img[0, 0, 1]) << 8)
            o = self.script.get(no, [])
            return [q for q, _t, _s in o], [(t, s_) for _q, t, s_ in o]
    for i in range(n):
        nf = int(rng.integers(1, 14))
        ocr = {}
        for no in range(1, nf + 1):
            items = []
            for _ in range(int(rng.integers(0, 4))):
                x0, y0 = int(rng.integers(0, W - 4)), int(rng.integers(0, H - 3))
                x1, y1 = x0 + int(rng.integers(1, 40)), y0 + int(rng.integers(1, 14))
                q = E.quad(x0, y0, x1, y1)
                if rng.random() < 0.3:
                    q = (np.asarray(q, np.float64) + rng.uniform(-1.5, 1.5, (4, 2))).tolist()
                text = str(rng.choice(["hello", "中文 mixed", "", " ", "tab\tin", "ｆｕｌｌ", "plain text", "日本語"]))
                items.append([q, text, round(float(rng.choice([rng.uniform(0.5, 1.0), 0.75, 0.7500001])), 7)])
            ocr[no] = items
        a0, b0 = sorted(int(v) for v in rng.integers(0, H, 2))
        c0, d0 = sorted(int(v) for v in rng.integers(0, W, 2))
        area = None if rng.random() < 0.25 else dict(ymin=a0, ymax=max(b0, a0 + 1), xmin=c0, xmax=max(d0, c0 + 1))
        tasks = []
        for no in rng.permutation(np.arange(1, nf + 3))[:int(rng.integers(1, nf + 3))].tolist():
            tasks.append([int(no), bool(rng.random() < 0.3) and no <= nf, None if area is not None or rng.random() < 0.5
                          else str(rng.choice(["LOWER_PART", "UPPER_PART", "UNKNOWN"]))])
        case = dict(n_frames=nf, lang=str(rng.choice(["ch", "en", "japan", "ch_tra"])), drop_score=float(rng.choice([0.75, 0.5, 0.0, 0.9])),
                    deviation=float(rng.choice([0.0, 0.0, 0.05, 0.3, 1.0])), area=area, tasks=tasks, ocr=ocr)
        try:
            ref, _seen = E.run_ocr_case(so, case)
        except Exception as e:                                      # noqa: BLE001
            ref = "EXC " + type(e).__name__ + str(e)[:80]
        try:
            src = extractor.ArraySource([frame_of(k + 1) for k in range(nf)], 25.0)
            tl = []
            for no, cached, default_area in tasks:
                o = ocr.get(no, [])
                dt, rr = ([q for q, _t, _s in o], [(t, s_) for _q, t, s_ in o]) if cached else (None, None)
                tl.append((nf, no, dt, rr, None, default_area))
            tl.append((nf, -1, None, None, None, None))
            ar = None if area is None else extractor.SubtitleArea(**area)
            mine = "".join(extractor.run_ocr_tasks(src, tl, Ocr(ocr), ar, case["lang"], case["drop_score"], case["deviation"], batch=3))
        except Exception as e:                                      # noqa: BLE001
            mine = "EXC " + type(e).__name__ + str(e)[:80]
        if ref != mine:
            bad += 1
            if bad <= 5:
                print("EXTRACT DIFF", {k: v for k, v in case.items() if k != "ocr"})
                print("  ocr :", str(ocr)[:700])
                print("  ref :", repr(ref)[:500])
                print("  mine:", repr(mine)[:500])
    print(f"extract: {n} cases, {bad} differences")
    return bad


def fuzz_fps(n, seed):
    """extract_frame_by_fps (backend/main.py:228-253) over random (frame count, fps, extract frequency)."""
    import make_frame_loop_golden as G
    from vse_amd import extractor
    main = G.install_stubs(80)
    rnd = random.Random(seed)
    bad = 0
    for i in range(n):
        n_frames = rnd.choice([0, 1, 2, 3, 7, 30, 31, 100, 257, rnd.randrange(0, 400)])
        fps = rnd.choice([23.976, 24.0, 25.0, 29.97, 30.0, 50.0, 59.94, 60.0, 12.0, 5.0, 1.0, 0.5, rnd.uniform(1, 120)])
        freq = rnd.choice([1, 2, 3, 3, 3, 5, 10, 24, 30, 60, 100])
        ext = object.__new__(main.SubtitleExtractor)
        ext.frame_count, ext.fps = n_frames, fps
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
        main.config.extractFrequency = G._Val(freq)
        main.config.subtitleArea = G._Val("AREA")
        ext.video_cap = Cap()
        ext.subtitle_ocr_task_queue = Q()
        ext.update_progress = lambda **k: None
        try:
            ext.extract_frame_by_fps()
            ref = tasks
        except Exception as e:                                      # noqa: BLE001
            ref = "EXC " + type(e).__name__
        try:
            mine = [[t[0], t[1], t[5]] for t in extractor.fps_tasks(n_frames, fps, freq, "AREA")]
        except Exception as e:                                      # noqa: BLE001
            mine = "EXC " + type(e).__name__
        if ref != mine:
            bad += 1
            if bad <= 5:
                print("FPS DIFF", n_frames, fps, freq, "ref", str(ref)[:200], "mine", str(mine)[:200])
    print(f"fps sampler: {n} cases, {bad} differences")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    bad = 0
    if a.only in ("", "srt"):
        bad += fuzz_srt(a.cases, a.seed)
    if a.only in ("", "filters"):
        bad += fuzz_filters(a.cases, a.seed)
    if a.only in ("", "cleanup"):
        bad += fuzz_cleanup(a.cases, a.seed)
    if a.only in ("", "loop"):
        bad += fuzz_frame_loop(a.cases, a.seed)
    if a.only in ("", "fps"):
        bad += fuzz_fps(a.cases, a.seed)
    # the two below install their own stub sets for the same module names: run them in separate invocations
    if a.only == "glue":
        bad += fuzz_glue(a.cases, a.seed)
    if a.only == "extract":
        bad += fuzz_extract(a.cases, a.seed)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
