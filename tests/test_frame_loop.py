"""Row a13 (accurate-mode frame loop): oracle and batched product selector against golden task lists produced by
the reference's own backend/main.py state machine (tests/golden/make_frame_loop_golden.py).  CPU only."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import frame_loop_ref as R

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "frame_loop.json")))


def _fakes(sc):
    calls = []

    def detect(no):
        return np.asarray(sc["frames"][no - 1]["boxes"], dtype=np.float32).reshape(-1, 4, 2)

    def predict(no):
        calls.append(no)
        o = sc["frames"][no - 1]["ocr"]
        return [q for q, _t, _s in o], [(t, s) for _q, t, s in o]
    return detect, predict, calls


def _norm(tasks):
    return [{"total": t[0], "frame_no": t[1], "cached": t[2] is not None,
             "texts": None if t[3] is None else [x[0] for x in t[3]]} for t in tasks]


def test_ratio_restatement():
    from vse_amd import frame_select
    for a, b, r in G["ratio_cases"]:
        assert abs(R.ratio(a, b) - r) < 1e-12 and abs(frame_select.similarity(a, b) - r) < 1e-12
    assert R.ratio("", "") == 1.0 and R.ratio("abc", "") == 0.0
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = "".join(rng.choice(list("abcd "), int(rng.integers(0, 12))))
        b = "".join(rng.choice(list("abcd "), int(rng.integers(0, 12))))
        assert abs(R.ratio(a, b) - frame_select.similarity(a, b)) < 1e-12


@pytest.mark.parametrize("k", range(len(G["scenarios"])))
def test_oracle_matches_reference_run(k):
    sc = G["scenarios"][k]
    detect, predict, calls = _fakes(sc)
    n = len(sc["frames"])
    tasks = R.extract_frame_by_det(range(1, n + 1), n, detect, predict, G["area"], G["threshold"])
    assert _norm(tasks) == sc["tasks"]
    assert calls == sc["predict_calls"]


@pytest.mark.parametrize("chunk", [1, 7, 64])
@pytest.mark.parametrize("prefetch", [False, True])
def test_batched_selector_matches_reference_run(chunk, prefetch):
    from vse_amd import frame_select
    area = SimpleNamespace(**G["area"])
    for sc in G["scenarios"]:
        detect, predict, calls = _fakes(sc)
        n = len(sc["frames"])
        sel = frame_select.AccurateFrameSelector(lambda fs: [detect(f) for f in fs], predict, area, n,
                                                 G["threshold"], chunk=chunk,
                                                 predict_batch=(lambda fs: [predict(f) for f in fs]) if prefetch else None)
        tasks = sel.run(range(1, n + 1))
        assert _norm(tasks) == sc["tasks"]
        if not prefetch:
            assert calls == sc["predict_calls"]          # same OCR invocations, in the same order
