#!/usr/bin/env python3
"""Generates tests/golden/ocr_glue.json by RUNNING the reference's own backend/tools/ocr.py in this container.

The reference module imports paddleocr / paddle / qfluentwidgets-based config, none of which is installed, so
those are replaced by inert stub modules; the code under test — OcrRecogniser.predict's box conversion, line
grouping and ordering (backend/tools/ocr.py:24-86), y_round (:16-22) and get_coordinates (:115-134) — is the
reference's, imported from /root/reference.  Only inputs and outputs are written out (data, not source).

Run once here: python tests/golden/make_ocr_glue_golden.py   (needs /root/reference; not run on the GPU box)
"""
import importlib
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ocr_glue.json")


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("paddleocr", PaddleOCR=object)
    mod("paddle")
    cfg = types.SimpleNamespace()
    mod("backend.config", config=cfg, tr={}, BASE_DIR=REF, __all__=["config", "tr", "BASE_DIR"])

    class _HA:
        onnx_providers = []

        def has_cuda(self):
            return False

    mod("backend.tools.hardware_accelerator", HardwareAccelerator=_HA)
    mod("backend.tools.paddle_model_config", PaddleModelConfig=object)
    pkg = mod("backend")
    pkg.__path__ = [os.path.join(REF, "backend")]
    tools = mod("backend.tools")
    tools.__path__ = [os.path.join(REF, "backend", "tools")]


def main():
    install_stubs()
    spec = importlib.util.spec_from_file_location("backend.tools.ocr", os.path.join(REF, "backend/tools/ocr.py"))
    ocr = importlib.util.module_from_spec(spec)
    sys.modules["backend.tools.ocr"] = ocr
    spec.loader.exec_module(ocr)

    rng = np.random.default_rng(20240929)
    cases = []

    def run_case(boxes, texts):
        rec = ocr.OcrRecogniser()
        rec.recogniser = lambda image, cls=False: (boxes, texts, None)
        dt, res = rec.predict(None)
        coords = ocr.get_coordinates(boxes)
        coords_nl = ocr.get_coordinates(np.asarray(boxes, dtype=np.float32).reshape(-1, 4, 2)) if len(boxes) else []
        cases.append({
            "boxes": [np.asarray(b, dtype=np.float64).tolist() for b in boxes],
            "texts": [[t, float(s)] for t, s in texts],
            "predict_boxes": [[list(map(int, p)) for p in b] for b in dt] if len(boxes) else [],
            "predict_res": [[t, float(s)] for t, s in res] if len(boxes) else [],
            "get_coordinates": [list(map(int, c)) for c in coords],
            "get_coordinates_ndarray": [list(map(int, c)) for c in coords_nl],
        })

    def quad(x0, y0, w, h, jitter=0.0):
        j = lambda: float(rng.uniform(-jitter, jitter))
        return np.array([[x0 + j(), y0 + j()], [x0 + w + j(), y0 + j()], [x0 + w + j(), y0 + h + j()],
                         [x0 + j(), y0 + h + j()]], dtype=np.float32)

    # empty
    run_case([], [])
    # hand-made: 3 boxes on 2 lines out of order; ties at y % 10 == 5 (y_round rounds 905 down, 906 up)
    run_case([quad(500, 905, 200, 40), quad(100, 906, 300, 40), quad(120, 960, 400, 42)],
             [("world", 0.98), ("hello", 0.99), ("line2", 0.97)])
    run_case([quad(10, 95, 50, 20), quad(200, 104, 50, 20), quad(100, 115, 50, 20), quad(300, 85, 40, 20)],
             [("a", 0.9), ("b", 0.8), ("c", 0.7), ("d", 0.6)])
    # random cases: 1-8 boxes, up to 3 lines, jittered quads, float coords
    for _ in range(60):
        nlines = int(rng.integers(1, 4))
        boxes, texts = [], []
        ys = sorted(rng.choice(np.arange(50, 1000, 1), size=nlines, replace=False).tolist())
        for li, y in enumerate(ys):
            for k in range(int(rng.integers(1, 4))):
                x = float(rng.integers(0, 1500))
                boxes.append(quad(x, y + float(rng.integers(-6, 7)), float(rng.integers(20, 400)),
                                  float(rng.integers(20, 70)), jitter=2.5))
                texts.append((f"t{li}_{k}", float(rng.uniform(0.3, 1.0))))
        perm = rng.permutation(len(boxes))
        run_case([boxes[i] for i in perm], [texts[i] for i in perm])
    yr = {str(y): int(ocr.OcrRecogniser.y_round(y)) for y in list(range(0, 40)) + [895, 900, 904, 905, 906, 909, 910, 1079]}
    with open(OUT, "w") as f:
        json.dump({"source": "backend/tools/ocr.py @ v2.2.0 executed with stubbed third-party imports",
                   "y_round": yr, "cases": cases}, f, separators=(",", ":"))
    print("wrote", OUT, len(cases), "cases")


if __name__ == "__main__":
    main()
