"""Sanity + known-answer checks of the CPU restatement (oracle/pipeline_ref.py).  CPU only."""
import numpy as np

from oracle import pipeline_ref as P


def test_det_resize_shape_examples():
    # 1080p, 4K and 720p all land on 544x960
    assert P.det_resize_shape(1080, 1920) == (544, 960)
    assert P.det_resize_shape(2160, 3840) == (544, 960)
    assert P.det_resize_shape(720, 1280) == (544, 960)
    assert P.det_resize_shape(360, 640) == (352, 640)
    assert P.det_resize_shape(10, 20) == (32, 32)


def test_cv2_resize_fixed_point_properties():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(P.cv2_resize_linear_u8(img, 53, 37), img)
    flat = np.full((40, 60, 3), 77, np.uint8)
    assert np.all(P.cv2_resize_linear_u8(flat, 31, 17) == 77)           # constants are preserved exactly
    # exact 2x downscale: fx = fy = 0.5 -> rounded mean of the 2x2 block (coefficients 1024/1024)
    big = rng.integers(0, 256, (8, 8, 1), dtype=np.uint8)
    out = P.cv2_resize_linear_u8(big, 4, 4)
    I = big.astype(np.int64)[..., 0]
    r0 = (I[0::2, 0::2] + I[0::2, 1::2]) * 1024
    r1 = (I[1::2, 0::2] + I[1::2, 1::2]) * 1024
    exp = (((1024 * (r0 >> 4)) >> 16) + ((1024 * (r1 >> 4)) >> 16) + 2) >> 2
    assert np.array_equal(out[..., 0], exp)


def test_db_postprocess_rectangle_known_answer():
    prob = np.zeros((544, 960), np.float32)
    prob[452:468, 191:706] = 0.9          # a 515x16 blob like the SURVEY §8(c) probe
    boxes, scores = P.db_postprocess(prob, 1080, 1920)
    assert len(boxes) == 1 and abs(scores[0] - 0.9) < 1e-6
    # min-area rect = [191..705]x[452..467]; unclip distance = A*1.5/L with A=514*15, L=2*(514+15)
    d = 514 * 15 * 1.5 / (2 * (514 + 15))
    x0, x1 = round(191 - d), round(705 + d)
    y0, y1 = round(452 - d), round(467 + d)
    exp = np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1]], np.float64)
    exp[:, 0] = np.round(exp[:, 0] / 960 * 1920)
    exp[:, 1] = np.round(exp[:, 1] / 544 * 1080)
    assert np.array_equal(boxes[0], exp.astype(np.float32)), (boxes[0], exp)


def test_db_postprocess_filters():
    prob = np.zeros((64, 96), np.float32)
    prob[10:12, 10:40] = 0.9              # short side 1 < min_size 3 -> dropped
    prob[30:40, 10:40] = 0.5              # score 0.5 < box_thresh 0.6 -> dropped
    prob[50:60, 50:90] = 0.25             # below bitmap threshold
    boxes, _ = P.db_postprocess(prob, 64, 96)
    assert len(boxes) == 0
    assert len(P.db_postprocess(np.zeros((32, 32), np.float32), 32, 32)[0]) == 0


def test_sorted_boxes_and_ctc():
    q = lambda x, y: np.array([[x, y], [x + 10, y], [x + 10, y + 5], [x, y + 5]], np.float32)
    bs = P.sorted_boxes([q(50, 100), q(10, 104), q(30, 20)])
    assert [tuple(b[0]) for b in bs] == [(30, 20), (10, 104), (50, 100)]
    probs = np.zeros((7, 5), np.float32)
    for t, c in enumerate([0, 2, 2, 0, 2, 3, 3]):
        probs[t, c] = 0.5 + 0.05 * t
    ids, conf = P.ctc_greedy(probs)
    assert ids == [2, 2, 3]
    assert abs(conf - np.mean([0.55, 0.7, 0.75])) < 1e-6
    assert P.ctc_greedy(np.eye(5, dtype=np.float32)[[0, 0, 0]]) == ([], 0.0)
    cs = P.en_charset()
    assert len(cs) == 97 and cs[1] == "0" and cs[0] == "blank" and cs[-1] == " "


def test_crop_identity_axis_aligned():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (60, 120, 3), dtype=np.uint8)
    quad = np.array([[10, 5], [90, 5], [90, 35], [10, 35]], np.float32)
    crop = P.get_rotate_crop_image(img, quad)
    assert crop.shape == (30, 80, 3)
    # integer-aligned, unscaled warp samples pixel centres exactly -> identical to slicing
    assert np.array_equal(crop, img[5:35, 10:90])
    tall = np.array([[10, 5], [20, 5], [20, 55], [10, 55]], np.float32)
    c2 = P.get_rotate_crop_image(img, tall)
    assert c2.shape == (10, 50, 3) and np.array_equal(c2, np.rot90(img[5:55, 10:20]))


def test_rec_batches_reference_grouping():
    crops = [np.zeros((48, w, 3), np.uint8) for w in (100, 900, 300, 50, 700, 480, 200)]
    groups = P.rec_batches(crops, 6)
    assert [g[0] for g in groups] == [[3, 0, 6, 2, 5, 4], [1]]
    assert groups[0][1] == int(48 * (700 / 48)) and groups[1][1] == 900
    x = P.resize_norm_img(crops[0], 320)
    assert x.shape == (3, 48, 320) and np.all(x[:, :, 100:] == 0) and np.all(x[:, :, :100] == -1.0)


def test_db_postprocess_hole_contours_known_answers():
    """cv2.findContours(RETR_LIST) also returns hole borders, and boxes_from_bitmap treats them like any contour.
    A 3x3 dip inside a 0.9 blob: border points = the 12 foreground pixels 4-adjacent to it (hull: an octagon inside
    [19,23]x[9,13]), min-area rectangle 4x4, score over its 25 lattice points (16 x 0.9 + 9 x dip) / 25."""
    prob = np.zeros((32, 64), np.float32)
    prob[4:21, 5:41] = 0.9
    prob[10:13, 20:23] = 0.25                       # score 0.666 >= 0.6: the hole yields a box of its own
    boxes, scores = P.db_postprocess(prob, 32, 64)
    assert len(boxes) == 2
    small = boxes[np.argmin([np.ptp(b[:, 0]) for b in boxes])]
    # 4x4 box grown by area * 1.5 / perimeter = 1.5 on every side: [17.5, 24.5] x [7.5, 14.5] -> integer corners
    assert small[:, 0].min() in (17, 18) and small[:, 0].max() in (24, 25)
    assert small[:, 1].min() in (7, 8) and small[:, 1].max() in (14, 15)
    assert abs(float(scores[np.argmin([np.ptp(b[:, 0]) for b in boxes])]) - (16 * 0.9 + 9 * 0.25) / 25) < 1e-6
    prob[10:13, 20:23] = 0.0                        # score 0.576 < 0.6: rejected
    assert len(P.db_postprocess(prob, 32, 64)[0]) == 1
    prob[10:13, 20:23] = 0.25
    prob[10:13, 5:20] = 0.25                        # the dip now reaches ... still enclosed by column 5? no: opens to x < 5
    assert len(P.db_postprocess(prob, 32, 64)[0]) == 1          # a notch open to the background is not a hole
    # a hole whose border pixels lie on the frame is still a hole (the frame itself is background for the scan)
    prob = np.zeros((16, 16), np.float32)
    prob[0:9, 0:9] = 0.9
    prob[3:6, 3:6] = 0.25
    assert len(P.db_postprocess(prob, 16, 16)[0]) == 2
    # two dips joined only diagonally are two holes (background is 4-connected)
    prob = np.zeros((32, 64), np.float32)
    prob[2:30, 2:60] = 0.95
    prob[10:13, 20:23] = 0.29
    prob[13:16, 23:26] = 0.29
    assert len(P.db_postprocess(prob, 32, 64)[0]) == 3
