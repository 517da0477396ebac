"""raw.txt -> SRT against vectors produced by the reference's own code
(tests/golden/make_srt_golden.py ran backend/main.py:614-637,731-864 with stubbed third-party imports)."""
import json
import os

import pytest

from vse_amd import srt

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "srt.json"), encoding="utf-8"))


@pytest.mark.parametrize("k", range(len(G["scenarios"])))
def test_srt_identical_to_reference(k, tmp_path):
    sc = G["scenarios"][k]
    tbl = {int(a): b for a, b in sc["msec"].items()} if "msec" in sc else None
    pos = (lambda no: float(tbl[no]) if no in tbl else None) if tbl is not None else (lambda no: None)
    text, short, norm = srt.generate_subtitle_file(sc["lines"], sc["fps"], sc["threshold"], pos)
    assert text == sc["srt"]
    assert short == sc["short"]
    assert "".join(norm) == sc["raw_after"]
    # file-based form: same side effects as the reference (raw.txt rewritten, SRT written)
    raw = tmp_path / "raw.txt"
    out = tmp_path / "out.srt"
    raw.write_text("".join(sc["lines"]), encoding="utf-8")
    assert srt.write_subtitle_file(str(raw), str(out), sc["fps"], sc["threshold"], pos) == sc["short"]
    assert out.read_text(encoding="utf-8") == sc["srt"] and raw.read_text(encoding="utf-8") == sc["raw_after"]


def test_ratio_definition():
    fl = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "frame_loop.json")))
    for a, b, r in fl["ratio_cases"]:
        assert srt.levenshtein_ratio(a, b) == pytest.approx(r, abs=1e-12)
    assert srt.levenshtein_ratio("", "") == 1.0 and srt.levenshtein_ratio("abcde", "abcdX") == pytest.approx(0.8)


def test_pipeline_lines_feed_the_writer():
    """extract_subtitles() raw lines (shim) -> SRT: frames with the same text collapse into one block."""
    from vse_amd import shim
    q = [(300, 850), (1500, 850), (1500, 920), (300, 920)]
    lines = []
    for no in range(1, 40):
        lines += shim.extract_subtitles(no, ([q], [("hello world", 0.99)]))
    for no in range(40, 90):
        lines += shim.extract_subtitles(no, ([q], [("another subtitle", 0.99)]))
    text, short, _ = srt.generate_subtitle_file(lines, 25.0)
    assert text.count("-->") == 2 and "hello world" in text and "another subtitle" in text and short == []
