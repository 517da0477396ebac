"""Oracle + host glue against the golden vectors produced by the reference's own backend/tools/ocr.py
(tests/golden/make_ocr_glue_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import pipeline_ref as P

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ocr_glue.json")))


def test_y_round_golden():
    for k, v in G["y_round"].items():
        assert P.y_round(int(k)) == v
    assert P.y_round(905) == 900 and P.y_round(906) == 910     # ties round down


@pytest.mark.parametrize("impl", ["oracle", "shim"])
def test_predict_glue_golden(impl):
    from vse_amd import shim
    for c in G["cases"]:
        boxes = [np.asarray(b, dtype=np.float32) for b in c["boxes"]]
        texts = [(t, s) for t, s in c["texts"]]
        if impl == "oracle":
            dt, res = P.ocr_predict_glue(boxes, texts)
            coords = P.get_coordinates(boxes)
            coords_nd = P.get_coordinates(np.asarray(boxes).reshape(-1, 4, 2)) if boxes else []
        else:
            dt, res = shim.OcrRecogniser.arrange(boxes, texts)
            coords = shim.get_coordinates(boxes)
            coords_nd = shim.get_coordinates(np.asarray(boxes).reshape(-1, 4, 2)) if boxes else []
        if boxes:
            assert [[list(p) for p in b] for b in dt] == c["predict_boxes"]
            assert [[t, s] for t, s in res] == c["predict_res"]
        else:
            assert len(dt) == 0 and len(res) == 0          # empty lists are passed through (ocr.py:85-86)
        assert [list(x) for x in coords] == c["get_coordinates"]
        assert [list(x) for x in coords_nd] == c["get_coordinates_ndarray"] == []   # non-list input -> []


def test_area_filter_matches_between_oracle_and_shim():
    from types import SimpleNamespace
    from vse_amd import shim
    rng = np.random.default_rng(0)
    area = (612, 717, 90, 1191)                 # Colab example (ymin,ymax,xmin,xmax), SURVEY §4
    ns = SimpleNamespace(ymin=612, ymax=717, xmin=90, xmax=1191)
    for _ in range(500):
        x0, y0 = int(rng.integers(0, 1200)), int(rng.integers(550, 760))
        c = (x0, x0 + int(rng.integers(1, 400)), y0, y0 + int(rng.integers(1, 80)))
        p = float(rng.uniform(0.5, 1))
        assert P.subtitle_area_keep(c, p, area) == shim.subtitle_area_keep(c, p, ns, 0, 0.75)
    assert P.subtitle_area_keep((100, 500, 620, 700), 0.9, area)          # fully inside, confident
    assert not P.subtitle_area_keep((100, 500, 620, 700), 0.75, area)     # strict > 0.75
    assert not P.subtitle_area_keep((100, 500, 600, 700), 0.9, area)      # sticks out of the area (rate 0)
    assert not P.subtitle_area_keep((100, 500, 10, 50), 0.9, area)        # no intersection


def test_model_selection_matrix_golden():
    """shim.PaddleModelConfig vs the reference's backend/tools/paddle_model_config.py run over all 88 languages x
    3 modes x accelerator on/off (tests/golden/make_model_config_golden.py), incl. its FileNotFoundError for 'kn'."""
    from vse_amd import shim
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "model_config.json")))

    class HA:
        def __init__(self, acc):
            self.acc = acc
            self.onnx_providers = []

        def has_accelerator(self):
            return self.acc
    old = (shim.config.language, shim.config.mode)
    try:
        for lang, mode, acc, det, rec, ver, shape, err in g["rows"]:
            shim.config.language, shim.config.mode = lang, mode
            if err is not None:
                with pytest.raises(FileNotFoundError):
                    shim.PaddleModelConfig(HA(acc))
                continue
            c = shim.PaddleModelConfig(HA(acc))
            assert (c.DET_MODEL_PATH, c.REC_MODEL_PATH, c.MODEL_VERSION, c.REC_IMAGE_SHAPE) == (det, rec, ver, shape), (lang, mode, acc)
    finally:
        shim.config.language, shim.config.mode = old
    assert len(g["rows"]) == 528


def test_raw_subtitle_line_format():
    from types import SimpleNamespace
    from vse_amd import shim
    boxes = [[(100, 850), (500, 850), (500, 900), (100, 900)], [(100, 10), (300, 10), (300, 40), (100, 40)]]
    res = [("hello\u4e16\u754c", 0.9), ("logo", 0.99)]
    area = SimpleNamespace(ymin=800, ymax=1000, xmin=50, xmax=1800)
    out = shim.extract_subtitles(42, (boxes, res), area, rec_char_type="en")
    assert out == ["00000042\t(100, 500, 850, 900)\thello\n"]          # CJK stripped for 'en', watermark outside
    out = shim.extract_subtitles(7, (boxes, res), None, rec_char_type="ch")
    assert out == ["00000007\t(100, 500, 850, 900)\thello\u4e16\u754c\n", "00000007\t(100, 300, 10, 40)\tlogo\n"]
    assert shim.extract_subtitles(1, ([], []), area) == []
