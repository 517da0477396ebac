"""Independent CONSISTENCY checks of the unpinned parts of the oracle.

These are NOT pins against the reference: paddlepaddle / paddleocr / cv2 exist on neither box, so the oracle's restatement of
their arithmetic (oracle/net_ref.py, oracle/pipeline_ref.py) cannot be executed side by side with them.  What can be done
cheaply is to check each restated primitive against an INDEPENDENT third-party or brute-force implementation of the same
published definition — a wrong gate order, weight layout, rounding rule or rectangle search shows up here:

  * `_lstm_ref` (Paddle `rnn` op, WeightList layout, gate order i,f,g,o)  vs  torch.nn.LSTM loaded with the same tensors
    (PyTorch's LSTM uses the same gate order and (w_ih, w_hh, b_ih, b_hh) parameterisation: cuDNN convention);
  * `cv2_resize_linear_u8` (OpenCV's 11-bit fixed-point INTER_LINEAR)     vs  a float64 half-pixel-centre bilinear resize:
    within +-1 grey level everywhere (the fixed-point error bound), exact on integer up/down-scales of constant images;
  * `_min_area_rect` (rotating over hull edges)                           vs  a brute-force search over 0.01-degree rotations:
    area never larger, and within 1e-3 relative of the brute-force minimum;
  * `_convex_hull`                                                        vs  scipy.spatial.ConvexHull vertex sets.
"""
import math

import numpy as np
import pytest
import torch

from oracle import net_ref
from oracle import pipeline_ref as P


@pytest.mark.parametrize("layers,bidirec,hidden,insz", [(1, False, 8, 5), (2, True, 16, 12), (2, True, 256, 96)])
def test_lstm_restatement_matches_torch_lstm(layers, bidirec, hidden, insz):
    torch.manual_seed(layers * 100 + hidden)
    ref = torch.nn.LSTM(insz, hidden, num_layers=layers, bidirectional=bidirec)
    ndir = 2 if bidirec else 1
    ws, bs = [], []
    for layer in range(layers):
        for d in range(ndir):
            sfx = f"_l{layer}" + ("_reverse" if d else "")
            ws += [getattr(ref, "weight_ih" + sfx).detach(), getattr(ref, "weight_hh" + sfx).detach()]
            bs += [getattr(ref, "bias_ih" + sfx).detach(), getattr(ref, "bias_hh" + sfx).detach()]
    x = torch.randn(11, 3, insz)
    with torch.no_grad():
        want, _ = ref(x)
    got = net_ref._lstm_ref(x, ws + bs, layers, bidirec, hidden)       # Paddle WeightList order: all weights, then all biases
    assert got.shape == want.shape
    assert float((got - want).abs().max()) < 2e-6


def _bilinear_f64(img, dst_w, dst_h):
    h, w = img.shape[:2]
    fy = np.clip((np.arange(dst_h) + 0.5) * (h / dst_h) - 0.5, 0, h - 1)
    fx = np.clip((np.arange(dst_w) + 0.5) * (w / dst_w) - 0.5, 0, w - 1)
    y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    wy, wx = (fy - y0)[:, None, None], (fx - x0)[None, :, None]
    I = img.astype(np.float64)
    top = I[y0][:, x0] * (1 - wx) + I[y0][:, x1] * wx
    bot = I[y1][:, x0] * (1 - wx) + I[y1][:, x1] * wx
    return top * (1 - wy) + bot * wy


@pytest.mark.parametrize("src,dst", [((1080, 1920), (544, 960)), ((720, 1280), (544, 960)), ((37, 53), (96, 160)),
                                     ((48, 211), (48, 320)), ((64, 64), (32, 32)), ((5, 7), (5, 7))])
def test_fixed_point_resize_is_within_one_level_of_float_bilinear(src, dst):
    rng = np.random.default_rng(src[0] * 7 + dst[1])
    img = rng.integers(0, 256, size=src + (3,), dtype=np.uint8)
    got = P.cv2_resize_linear_u8(img, dst[1], dst[0]).astype(np.float64)
    want = _bilinear_f64(img, dst[1], dst[0])
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1.0 + 1e-9          # 11-bit coefficients + two truncating shifts: at most one level
    # the two truncating shifts (>> 4, >> 16) ahead of the rounding (+ 2) >> 2 leave OpenCV's known small negative bias
    assert abs(float((got - want).mean())) < 0.25          # (exact 2:1 averages land on .25 / .5 / .75 and round half up: small positive bias there)
    const = np.full(src + (3,), 173, np.uint8)
    assert (P.cv2_resize_linear_u8(const, dst[1], dst[0]) == 173).all()


def _brute_min_area(pts, step_deg=0.01):
    best = None
    P_ = np.asarray(pts, np.float64)
    for k in range(int(round(90.0 / step_deg))):
        a = math.radians(k * step_deg)
        c, s = math.cos(a), math.sin(a)
        u = P_[:, 0] * c + P_[:, 1] * s
        v = -P_[:, 0] * s + P_[:, 1] * c
        area = (u.max() - u.min()) * (v.max() - v.min())
        if best is None or area < best:
            best = area
    return best


def test_min_area_rect_against_brute_force_rotation_search():
    rng = np.random.default_rng(5)
    for case in range(40):
        n = int(rng.integers(3, 40))
        if case % 3 == 0:      # a rotated text-like blob: long thin cloud
            ang = rng.uniform(0, math.pi)
            base = np.stack([rng.uniform(-80, 80, n), rng.uniform(-9, 9, n)], 1)
            rot = np.array([[math.cos(ang), -math.sin(ang)], [math.sin(ang), math.cos(ang)]])
            pts = np.rint(base @ rot.T + 200).astype(np.int64)
        else:
            pts = rng.integers(0, 300, size=(n, 2))
        hull = P._convex_hull(pts)
        if len(hull) < 3:
            continue
        corners, w, h = P._min_area_rect(hull)
        area = w * h
        brute = _brute_min_area(hull)
        assert area <= brute * (1 + 1e-9), (case, area, brute)            # an edge-aligned rectangle is optimal (rotating calipers)
        assert area >= brute * (1 - 1e-3), (case, area, brute)            # and the brute-force grid gets within 1e-3 of it
        # the returned corners span that rectangle and contain every hull point
        e0, e1 = corners[1] - corners[0], corners[3] - corners[0]
        assert abs(np.linalg.norm(e0) * np.linalg.norm(e1) - area) <= 1e-6 * max(area, 1.0)
        assert abs(float(e0 @ e1)) <= 1e-6 * max(area, 1.0)
        rel = np.asarray(hull, np.float64) - corners[0]
        for e in (e0, e1):
            t = rel @ e / float(e @ e)
            assert t.min() >= -1e-9 and t.max() <= 1 + 1e-9


def test_convex_hull_against_scipy():
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(9)
    for _ in range(30):
        pts = rng.integers(0, 60, size=(int(rng.integers(4, 80)), 2))
        if np.linalg.matrix_rank(pts - pts[0]) < 2:
            continue
        ours = set(map(tuple, P._convex_hull(pts)))
        sp = ConvexHull(pts.astype(np.float64))
        theirs = set(map(tuple, pts[sp.vertices]))
        assert ours == theirs          # strict hull: collinear boundary points dropped by both


@pytest.mark.parametrize("mid", ["V4_ch_rec", "V4_ch_rec_fast", "V4_en_rec_fast", "V3_ch_rec_fast", "V4_ch_det", "V4_ch_det_fast", "V2_ch_det"])
def test_standin_weights_are_alive_and_well_conditioned(mid):
    """The seeded stand-in weights (net_ref.synth_weights + calibrate) must make nets that (a) ANSWER DIFFERENT INPUTS DIFFERENTLY — until
    round 5 the PP-LCNetV3 stand-ins were constant functions of their input (their "learnable affine" scales were drawn as N(0, 0.05))
    and the deep HGNet ones nearly so (layer-wide LSUV scaling lets per-channel offsets swamp the input-dependent part), which made every
    parity test on them a test of bias propagation — and (b) are not chaotic: rounding the weights to fp16 alone must move the outputs
    by far less than two inputs differ.  Consistency checks of test infrastructure, not pins."""
    desc, w = net_ref.get_weights(mid)
    det = "_det" in mid
    shape = (2, 3, 64, 96) if det else (2, 3, 48, 160)
    rng = np.random.default_rng(0)
    xa = rng.uniform(-1, 1, shape).astype(np.float16).astype(np.float32)
    xb = rng.uniform(-1, 1, shape).astype(np.float16).astype(np.float32)
    w16 = {k: (v.astype(np.float16).astype(np.float32) if v.ndim >= 2 else v) for k, v in w.items()}
    pa = net_ref.run_graph(desc, w, xa)[0].numpy().astype(np.float64)
    pb = net_ref.run_graph(desc, w, xb)[0].numpy().astype(np.float64)
    p16 = net_ref.run_graph(desc, w16, xa)[0].numpy().astype(np.float64)
    if det:
        assert pa.max() - pa.min() > 0.5 and 0.05 < float((pa > 0.3).mean()) < 0.95          # a map that crosses the DB threshold
        between, rounding = float(np.abs(pa - pb).mean()), float(np.abs(pa - p16).max())
        assert between > 0.02 and rounding < 5e-3 and rounding < 0.1 * between, (mid, between, rounding)
    else:
        lg = lambda p: np.log(np.maximum(p, 1e-300))
        between = float(np.median(np.abs(lg(pa) - lg(pb))))
        rounding = np.abs(lg(pa) - lg(p16))
        assert len(set(pa.argmax(-1).ravel().tolist())) >= 6, "the arg-max must move along the sequence"
        assert between > 0.3 and float(rounding.max()) < 0.2 and float(np.median(rounding)) < 0.05 * between, (mid, between, float(rounding.max()))
