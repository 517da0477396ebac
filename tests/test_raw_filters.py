"""raw.txt watermark / scene-text filters against vectors produced by the reference's own code
(tests/golden/make_raw_filters_golden.py ran backend/main.py:506-612,671-729,866-881 with scripted answers)."""
import json
import os

import pytest

from vse_amd import raw_filters as F

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "raw_filters.json"), encoding="utf-8"))


def _lines(text):
    return text.splitlines(keepends=True)


@pytest.mark.parametrize("k", range(len(G["scenarios"])))
def test_filters_identical_to_reference(k, tmp_path):
    sc = G["scenarios"][k]
    areas, out = F.detect_watermark_area(sc["lines"])
    assert [[list(a), c] for a, c in areas] == sc["watermark_areas"]
    assert "".join(out) == sc["raw_after_detect"]
    assert [[list(a), c] for a, c in F.detect_subtitle_area(_lines(sc["raw_after_detect"]))] == sc["subtitle_area"]

    answers = list(sc["answers"])
    asked = []

    def decide(x):
        asked.append(x)
        return answers.pop(0) == "y"
    raw = tmp_path / "raw.txt"
    raw.write_text("".join(sc["lines"]), encoding="utf-8")
    F.filter_watermark_file(str(raw), decide)
    assert len(asked) == sc["n_watermark_questions"]
    assert raw.read_text(encoding="utf-8") == sc["raw_after_watermark"]
    if sc["scene_text_error"]:
        with pytest.raises(IndexError):
            F.filter_scene_text_file(str(raw), decide)
    else:
        F.filter_scene_text_file(str(raw), decide)
    assert raw.read_text(encoding="utf-8") == sc["raw_after_scene_text"]


def _unite_reference_walk(coords, tx, ty):
    """The double loop as the reference writes it (main.py:875-881), for randomised cross-checks of the numpy walk."""
    def similar(a, b):
        return abs(a[0] - b[0]) < tx and abs(a[1] - b[1]) < tx and abs(a[2] - b[2]) < ty and abs(a[3] - b[3]) < ty
    index = 0
    for c in coords:
        for i in coords:
            if similar(c, i):
                coords[index] = i
        index += 1
    return coords


def test_unite_coordinates_matches_the_plain_walk():
    import random
    rnd = random.Random(3)
    for trial in range(40):
        n = rnd.randint(0, 60)
        tx, ty = rnd.choice([(100, 50), (1, 1), (30, 200), (0, 5)])
        coords = [(rnd.randint(0, 400), rnd.randint(0, 400), rnd.randint(0, 200), rnd.randint(0, 200)) for _ in range(n)]
        assert F.unite_coordinates(list(coords), tx, ty) == _unite_reference_walk(list(coords), tx, ty)


def test_filters_feed_the_srt_writer():
    from vse_amd import srt
    lines = [f"{n:08d}\t(300, 1500, 850, 920)\thello\n" for n in range(1, 40)] + [f"{n:08d}\t(1700, 1850, 40, 90)\tLOGO\n" for n in range(1, 40)]
    lines = F.filter_watermark(lines, lambda a: a[0] == (1700, 1850, 40, 90))
    lines = F.filter_scene_text(lines)
    text, _short, _ = srt.generate_subtitle_file(lines, 25.0)
    assert "LOGO" not in text and text.count("-->") == 1
