"""SRT text clean-up (vse_amd.text_cleanup, the rest of row N4) against tests/golden/text_cleanup.json: the reference's own
backend/tools/reformat.py executed on scripted SRT files with a scripted word segmenter (make_text_cleanup_golden.py)."""
import json
import os
import re

import pytest

from vse_amd import text_cleanup

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "text_cleanup.json"), encoding="utf-8"))
VOCAB = G["vocab"]


def scripted_segment(text):
    s = re.sub("[^a-z0-9]", "", text.lower())
    out, i = [], 0
    while i < len(s):
        for w in VOCAB:
            if s.startswith(w, i):
                out.append(w)
                i += len(w)
                break
        else:
            out.append(s[i])
            i += 1
    return out


@pytest.mark.parametrize("k", range(len(G["cases"])))
def test_files_match_reference(k, tmp_path):
    c = G["cases"][k]
    assert c["ok"]
    p = tmp_path / "x.srt"
    p.write_text(c["input"], encoding="utf-8")
    text_cleanup.execute(str(p), c["lang"], segment=scripted_segment, typo_map=G["typo_map"])
    assert p.read_text(encoding="utf-8") == c["output"]


def test_default_typo_map_is_the_references():
    assert text_cleanup.DEFAULT_TYPO_MAP == G["typo_map"]


def test_missing_segmenter_is_an_error():
    import importlib.util
    if importlib.util.find_spec("wordsegment") is not None:
        pytest.skip("wordsegment is installed here")
    with pytest.raises(RuntimeError, match="word segmenter"):
        text_cleanup.execute("/nonexistent.srt", "en")


def test_a_failing_block_is_kept_and_execute_returns_true(tmp_path):
    """reformat.execute (backend/tools/reformat.py:16-214) keeps a block whose clean-up raises (segmenter failure, a segment
    with regex metacharacters) and goes on with the next one; it returns True."""
    srt = "1\n00:00:01,000 --> 00:00:02,000\nhelloworld (a+b)*[c\n\n2\n00:00:03,000 --> 00:00:04,000\nthisisfine\n\n"
    p = tmp_path / "x.srt"
    p.write_text(srt, encoding="utf-8")

    def seg(text):
        if "(" in text or "hello" in text:
            raise RuntimeError("segmenter failure")
        return ["this", "is", "fine"]
    assert text_cleanup.execute(str(p), "en", segment=seg, typo_map={}) is True
    out = p.read_text(encoding="utf-8")
    assert "helloworld (a+b)*[c" in out and "this is fine" in out.lower()


def test_sharded_run_without_process_group_raises():
    from vse_amd import extractor

    class Ocr:
        def predict(self, f):
            return [], []
    import numpy as np
    src = extractor.ArraySource(np.zeros((4, 8, 8, 3), np.uint8), 25.0)
    tasks = [(2, 1, None, None, None, None), (2, 2, None, None, None, None)]
    with pytest.raises(RuntimeError, match="not initialised"):
        extractor.run_ocr_tasks(src, tasks, Ocr(), None, "ch", 0.5, 0.0, shard=(0, 2))
