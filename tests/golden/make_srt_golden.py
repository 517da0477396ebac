#!/usr/bin/env python3
"""Generates tests/golden/srt.json by RUNNING the reference's own raw.txt -> SRT code
(SubtitleExtractor._concat_content_with_same_frameno, _remove_duplicate_subtitle, generate_subtitle_file,
_frame_to_timecode; backend/main.py:614-637,731-864) in this container on scripted raw subtitle files.

Same stub set as make_frame_loop_golden.py (imported from it).  Stubs that carry behaviour and are therefore part of
what the vectors pin:
  * Levenshtein.ratio -> normalised InDel similarity 2*LCS/(len a + len b) (Levenshtein==0.26.0, requirements.txt:2);
  * cv2.VideoCapture -> a scripted capture: per scenario either "no container timestamps" (read() fails, the
    frame-rate fallback of _frame_to_timecode is taken) or a table frame_no -> CAP_PROP_POS_MSEC.
Only inputs and outputs are written (data, not source).  Needs /root/reference; not run on the GPU box.
"""
import json
import os
import sys
import tempfile
import types

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_frame_loop_golden as G  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "srt.json")


def raw(frame_no, text, coord=(300, 1500, 850, 920)):
    return f"{str(frame_no).zfill(8)}\t{coord}\t{text}\n"


def make_scenarios():
    S = []
    # 1: two subtitles, durations > 1 s at 25 fps
    S.append(dict(fps=25.0, lines=[raw(n, "hello world") for n in range(10, 60)] + [raw(n, "second line") for n in range(70, 130)]))
    # 2: OCR noise inside one subtitle: the LONGEST variant is kept; spaces are ignored by the similarity
    S.append(dict(fps=25.0, lines=[raw(10, "the quick brown fox"), raw(11, "the quick brown f0x jumps"), raw(12, "thequick brown fox"),
                                   raw(40, "the quick brown fox"), raw(41, "completely different")]))
    # 3: single-frame subtitles (end = next line's start), last line single
    S.append(dict(fps=30.0, lines=[raw(5, "a"), raw(9, "bbbb"), raw(50, "cccccc")]))
    # 4: shorter than one second -> shown for fps frames; returned as post-process list
    S.append(dict(fps=24.0, lines=[raw(100, "short one"), raw(101, "short one"), raw(200, "long one")] + [raw(n, "long one") for n in range(201, 260)]))
    # 5: several lines on the same frame are concatenated with a space (in file order), NFKC applied
    S.append(dict(fps=25.0, lines=[raw(10, "top line", (300, 900, 800, 840)), raw(10, "bottom line"), raw(11, "top line", (300, 900, 800, 840)),
                                   raw(11, "bottom line"), raw(12, "ｆｕｌｌ　ｗｉｄｔｈ １２３"), raw(13, "ｆｕｌｌ　ｗｉｄｔｈ １２３"),
                                   raw(60, "ﬁ ligature ①"), raw(60, "②"), raw(60, "third")]))
    # 6: empty texts and whitespace-only texts
    S.append(dict(fps=25.0, lines=[raw(1, ""), raw(2, ""), raw(3, " "), raw(30, "x"), raw(31, "")]))
    # 7: similarity exactly at / around the threshold (0.8): 'abcde' vs 'abcdX' = 0.8 -> NOT < 0.8 -> same subtitle
    S.append(dict(fps=25.0, lines=[raw(1, "abcde"), raw(2, "abcdX"), raw(3, "abcXY"), raw(4, "abXYZ"), raw(40, "abXYZ")]))
    # 8: container timestamps available (variable frame rate): msec table
    tbl = {n: round(n * 41.7083) for n in range(0, 400)}
    S.append(dict(fps=23.976, msec=tbl, lines=[raw(n, "with timestamps") for n in range(24, 100)] + [raw(150, "tail")]))
    # 9: timestamps present but zero for frame 0 (milliseconds <= 0 -> fallback formula), hours/minutes roll-over
    tbl = {0: 0, 1: 40, 90000: 3600000, 90100: 3604000, 1500: 60000, 1600: 64000}
    S.append(dict(fps=25.0, msec=tbl, lines=[raw(0, "first"), raw(1, "first"), raw(1500, "minute mark")] + [raw(n, "minute mark") for n in (1600,)] +
                  [raw(90000, "hour mark"), raw(90100, "hour mark")]))
    # 10: empty raw file
    S.append(dict(fps=25.0, lines=[]))
    # 11: one line only
    S.append(dict(fps=25.0, lines=[raw(7, "only")]))
    # 12: Chinese + mixed, lower threshold
    S.append(dict(fps=25.0, threshold=60, lines=[raw(10, "你好，世界"), raw(11, "你好,世界!"), raw(12, "你好世界"), raw(50, "再见"), raw(51, "再 见")]))
    return S


def run_reference(main, sc):
    class Cap:
        def __init__(self, path):
            self.no = None

        def set(self, prop, v):
            self.no = int(v)

        def read(self):
            t = sc.get("msec")
            return (t is not None and self.no in t), None

        def get(self, prop):
            return float(sc["msec"][self.no])

        def release(self):
            pass
    main.cv2.VideoCapture = Cap
    main.cv2.CAP_PROP_POS_FRAMES = 1
    main.cv2.CAP_PROP_POS_MSEC = 0
    main.config.thresholdTextSimilarity = G._Val(sc.get("threshold", 80))
    with tempfile.TemporaryDirectory() as d:
        ext = object.__new__(main.SubtitleExtractor)
        ext.raw_subtitle_path = os.path.join(d, "raw.txt")
        ext.subtitle_output_path = os.path.join(d, "out.srt")
        ext.video_path = "video.mp4"
        ext.fps = sc["fps"]
        ext.use_vsf = False
        ext.append_output = lambda *a, **k: None
        with open(ext.raw_subtitle_path, "w", encoding="utf-8") as f:
            f.writelines(sc["lines"])
        short = ext.generate_subtitle_file()
        with open(ext.subtitle_output_path, encoding="utf-8") as f:
            srt = f.read()
        with open(ext.raw_subtitle_path, encoding="utf-8") as f:
            raw_after = f.read()
    return srt, short, raw_after


def main():
    m = G.install_stubs(80)
    import collections
    m.tr = collections.defaultdict(lambda: collections.defaultdict(lambda: "{}"))
    out = {"source": "backend/main.py:614-637,731-864 @ v2.2.0 executed with stubbed third-party imports", "scenarios": []}
    for sc in make_scenarios():
        srt, short, raw_after = run_reference(m, sc)
        rec = {"fps": sc["fps"], "threshold": sc.get("threshold", 80), "lines": sc["lines"], "srt": srt, "short": short,
               "raw_after": raw_after}
        if "msec" in sc:
            rec["msec"] = {str(k): v for k, v in sc["msec"].items()}
        out["scenarios"].append(rec)
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, separators=(",", ":"))
    print("wrote", OUT, len(out["scenarios"]), "scenarios")


if __name__ == "__main__":
    main()
