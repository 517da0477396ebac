"""GPU pre/post-processing kernels vs the CPU oracle: integer/byte work is compared bit-exactly."""
import numpy as np
import pytest

from oracle import net_ref
from oracle import pipeline_ref as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hw", [(1080, 1920), (720, 1280), (360, 640), (544, 960), (97, 131)])
def test_det_preprocess_bit_exact(ctx, hw):
    import torch
    rng = np.random.default_rng(hw[0])
    n = 2
    frames = rng.integers(0, 256, (n,) + hw + (3,), dtype=np.uint8)
    rh, rw = P.det_resize_shape(*hw)
    got = ctx.det_preprocess(torch.from_numpy(frames).cuda(), rh, rw).cpu().numpy()
    assert got.shape == (n, rh, rw, 8) and np.all(got[..., 3:] == 0)
    for f in range(n):
        x, _ = P.det_preprocess(frames[f])
        ref = x[0].transpose(1, 2, 0).astype(np.float16)          # the engine stores fp16
        assert np.array_equal(got[f, ..., :3], ref)


def test_det_preprocess_row_pitched_view(ctx):
    """The reference passes sliced views (frame[cropped:], subtitle_ocr.py:283): non-owning, offset start."""
    import torch
    rng = np.random.default_rng(9)
    frames = rng.integers(0, 256, (1, 200, 320, 3), dtype=np.uint8)
    dev = torch.from_numpy(frames).cuda()
    view = dev[:, 100:]
    got = ctx.det_preprocess(view, 96, 320).cpu().numpy()
    ref, _ = P.det_preprocess(frames[0, 100:])
    assert ref.shape[2:] == (96, 320)
    assert np.array_equal(got[0, ..., :3], ref[0].transpose(1, 2, 0).astype(np.float16))


def _synthetic_maps(seed, n=3, h=160, w=256):
    rng = np.random.default_rng(seed)
    maps = np.zeros((n, h, w), np.float32)
    for f in range(n):
        for _ in range(int(rng.integers(1, 6))):
            x0, y0 = int(rng.integers(5, w - 80)), int(rng.integers(5, h - 30))
            bw, bh = int(rng.integers(8, 70)), int(rng.integers(3, 20))
            ang = rng.uniform(-0.3, 0.3)
            ys, xs = np.mgrid[0:h, 0:w]
            u = (xs - x0) * np.cos(ang) + (ys - y0) * np.sin(ang)
            v = -(xs - x0) * np.sin(ang) + (ys - y0) * np.cos(ang)
            m = (u >= 0) & (u < bw) & (v >= 0) & (v < bh)
            maps[f][m] = np.maximum(maps[f][m], rng.uniform(0.35, 0.99))
        maps[f] += rng.uniform(0, 0.25, (h, w)).astype(np.float32) * (maps[f] == 0)
    return maps


@pytest.mark.parametrize("w", [1024, 1027, 2500, 3840])
def test_db_labels_across_row_segments(ctx, w):
    """The label passes give a block one 1024-pixel SEGMENT of a row and start every text pixel on its run's first pixel inside that
    segment (round 5); runs that cross a segment seam (x = 1024, 2048, ...), end on one, start on one, touch the map's last column or a
    width that is not a multiple of 4 must still come out as ONE component each: boxes identical to the oracle's."""
    import torch
    h = 48
    rng = np.random.default_rng(w)
    maps = np.zeros((3, h, w), np.float32)
    for f in range(3):
        bars = [(4, 900, min(w, 1200)), (12, 1020, min(w, 1024)), (20, min(w - 40, 1024), w), (28, 8, w - 3), (36, 2040 % (w - 60), 2040 % (w - 60) + 40)]
        for y0, xa, xb in bars:
            if xb - xa < 6:
                continue
            maps[f, y0 + f:y0 + f + 5, xa:xb] = rng.uniform(0.7, 0.95)
        maps[f, 2:46:9, 1023 % w] = 0.9                      # single pixels on a seam column
        maps[f] += rng.uniform(0, 0.2, (h, w)).astype(np.float32) * (maps[f] == 0)
    got = ctx.db_postprocess(torch.from_numpy(maps).cuda(), 2 * h, 2 * w)
    total = 0
    for f in range(3):
        rb, rs = P.db_postprocess(maps[f], 2 * h, 2 * w)
        gb, gs = got[f]
        assert gb.shape == rb.shape and np.array_equal(gb, rb), (w, f, gb, rb)
        assert np.abs(gs - rs).max() < 1e-6 if len(rs) else True
        total += len(rb)
    assert total >= 9


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_db_postprocess_matches_oracle(ctx, seed):
    import torch
    maps = _synthetic_maps(seed)
    n, h, w = maps.shape
    src_h, src_w = 2 * h + 7, 2 * w + 3
    got = ctx.db_postprocess(torch.from_numpy(maps).cuda(), src_h, src_w)
    total = 0
    for f in range(n):
        rb, rs = P.db_postprocess(maps[f], src_h, src_w)
        gb, gs = got[f]
        assert gb.shape == rb.shape, (f, gb, rb)
        assert np.array_equal(gb, rb), (f, gb, rb)                 # integer pixel coordinates: exact
        assert np.abs(gs - rs).max() < 1e-6 if len(rs) else True
        total += len(rb)
    assert total > 0


def test_db_postprocess_on_real_detector_map(ctx):
    import torch
    from vse_amd import synth
    frames = synth.make_frames(2, 540, 960, seed=11)
    desc, w = net_ref.get_weights("V3_ch_det_fast")
    maps = []
    for f in frames:
        x, _ = P.det_preprocess(f)
        maps.append(net_ref.run_graph(desc, w, x)[0].numpy()[0, 0])
    maps = np.stack(maps)
    got = ctx.db_postprocess(torch.from_numpy(maps).cuda(), 540, 960)
    nb = 0
    for f in range(2):
        rb, rs = P.db_postprocess(maps[f], 540, 960)
        assert np.array_equal(got[f][0], rb)
        nb += len(rb)
    assert nb >= 2
    empty = ctx.db_postprocess(torch.zeros((1, 64, 64), device="cuda"), 64, 64)
    assert empty[0][0].shape == (0, 4, 2)


def test_rec_preprocess_matches_oracle(ctx):
    import torch
    from vse_amd import pipeline, synth
    frames, truth = synth.make_frames(2, 540, 960, seed=21, return_truth=True)
    dev = torch.from_numpy(frames).cuda()
    quads, specs = [], []
    rng = np.random.default_rng(0)
    for f, tr in enumerate(truth):
        for (x0, y0, x1, y1, _t) in tr:
            j = rng.uniform(-2, 2, (4, 2))
            q = np.array([[x0, y0], [x1, y0 + 3], [x1, y1 + 3], [x0, y1]], np.float32) + j.astype(np.float32)
            quads.append((f, q))
    quads.append((0, np.array([[100, 50], [130, 50], [130, 200], [100, 200]], np.float32)))     # tall -> rot90
    img_w = 640
    for f, q in quads:
        cw, ch, rot = pipeline.crop_geometry(q)
        iw, ih = (ch, cw) if rot else (cw, ch)
        rw = P.rec_resized_width(iw, ih, img_w)
        specs.append(dict(quad=q, frame=f, crop_w=cw, crop_h=ch, resized_w=rw, rotate=rot))
    got = ctx.rec_preprocess(dev, specs, 48, img_w).cpu().numpy()
    for k, (f, q) in enumerate(quads):
        crop = P.get_rotate_crop_image(frames[f], q)
        ref = P.resize_norm_img(crop, img_w).transpose(1, 2, 0)
        # integer arithmetic end to end (1/32-pixel coordinates, cv2's 15-bit bicubic weight table, fixed-point resize): bit-exact
        assert np.array_equal(got[k, ..., :3], ref.astype(np.float16)), (k, np.abs(got[k, ..., :3].astype(np.float32) - ref).max())
        assert np.all(got[k, :, specs[k]["resized_w"]:, :] == 0)


def test_ctc_collapse_matches_oracle(ctx):
    import torch
    rng = np.random.default_rng(4)
    b, t, c = 37, 150, 40
    probs = rng.dirichlet(np.ones(c) * 0.3, size=(b, t)).astype(np.float32)
    probs[0] = 0
    probs[0, :, 0] = 1.0                                        # all blank
    probs[1] = 0
    probs[1, :, 5] = 1.0                                        # one long run -> a single symbol
    idx = probs.argmax(-1).astype(np.int32)
    mp = probs.max(-1).astype(np.float32)
    pairs = np.stack([idx.view(np.float32), mp], -1).reshape(b, 1, t, 2)
    oi, ol, oc = ctx.ctc_collapse(torch.from_numpy(np.ascontiguousarray(pairs)).cuda())
    oi, ol, oc = oi.cpu().numpy(), ol.cpu().numpy(), oc.cpu().numpy()
    for r in range(b):
        ids, conf = P.ctc_greedy(probs[r])
        assert oi[r, :ol[r]].tolist() == ids
        assert abs(oc[r] - conf) < 1e-5
    assert ol[0] == 0 and oc[0] == 0.0 and ol[1] == 1


def test_db_postprocess_noise_frame_degrades_alone(ctx):
    """A frame whose probability map is noise-like (more run end-points than the device record buffer holds: TV static, a
    detector gone wild) must not abort the batch: it alone takes the host pass, gives the oracle's boxes (up to
    max_candidates, like the reference), and the other frames are untouched."""
    import torch
    from oracle import pipeline_ref as P
    rng = np.random.default_rng(3)
    h, w = 544, 960
    clean = np.zeros((h, w), np.float32)
    clean[400:440, 200:700] = 0.9
    # ~4000 isolated 6x6 blobs = ~49k run end-points (> 32768 device records) and more components than max_candidates
    cells = rng.random((h // 8, w // 8)) < 0.5
    noise = np.zeros((h, w), np.float32)
    for cy, cx in zip(*np.nonzero(cells)):
        noise[cy * 8 + 1:cy * 8 + 7, cx * 8 + 1:cx * 8 + 7] = 0.9
    prob = np.stack([clean, noise, clean])
    res = ctx.db_postprocess(torch.from_numpy(prob).cuda(), 1080, 1920, max_boxes=8192)
    for f in (0, 2):
        rb, _ = P.db_postprocess(prob[f], 1080, 1920)
        assert len(res[f][0]) == len(rb) == 1 and np.array_equal(res[f][0], np.asarray(rb, np.float32))
    rb, rs = P.db_postprocess(prob[1], 1080, 1920)
    assert len(res[1][0]) == len(rb) == 1000          # max_candidates, like the reference
    key = lambda b: tuple(np.asarray(b).reshape(-1).tolist())
    assert sorted(map(key, res[1][0])) == sorted(map(key, rb))


def _holey_maps(seed, n=4, h=96, w=160):
    """High-probability blobs with dips below the threshold: enclosed holes of several sizes and depths (some pass the
    0.6 box score, some do not), islands inside holes, notches open to the background, diagonally touching holes, blobs on
    the frame."""
    rng = np.random.default_rng(seed)
    maps = rng.uniform(0, 0.2, (n, h, w)).astype(np.float32)
    for f in range(n):
        for _ in range(int(rng.integers(2, 6))):
            bw, bh = int(rng.integers(20, 70)), int(rng.integers(10, 30))
            x0, y0 = int(rng.integers(-5, w - bw + 5)), int(rng.integers(-3, h - bh + 3))
            xa, ya, xb, yb = max(x0, 0), max(y0, 0), min(x0 + bw, w), min(y0 + bh, h)
            maps[f, ya:yb, xa:xb] = rng.uniform(0.7, 0.99)
            for _ in range(int(rng.integers(1, 5))):
                dw, dh = int(rng.integers(1, 7)), int(rng.integers(1, 6))
                dx, dy = int(rng.integers(xa - 2, max(xa - 1, xb - dw + 2))), int(rng.integers(ya - 2, max(ya - 1, yb - dh + 2)))
                dxa, dya = max(dx, 0), max(dy, 0)
                maps[f, dya:max(dya, dy + dh), dxa:max(dxa, dx + dw)] = rng.choice([0.0, 0.2, 0.29, 0.3])
                if dw >= 3 and dh >= 3 and rng.random() < 0.5:
                    maps[f, dy + 1:dy + 2, dx + 1:dx + 2] = 0.95            # an island inside the hole
                if rng.random() < 0.4:                                     # a second dip touching the first one diagonally
                    maps[f, max(dy + dh, 0):max(dy + dh, 0) + 2, max(dx + dw, 0):max(dx + dw, 0) + 2] = 0.1
    return maps


@pytest.mark.parametrize("seed", [10, 11, 12])
def test_db_postprocess_hole_contours_match_oracle(ctx, seed):
    """RETR_LIST semantics: hole borders are contours too.  The engine finds them from the run records on the host
    (db_geometry.h hole_borders: union of background runs), the oracle by labelling the inverted mask with scipy."""
    import torch
    maps = _holey_maps(seed)
    n, h, w = maps.shape
    got = ctx.db_postprocess(torch.from_numpy(maps).cuda(), 2 * h, 2 * w)
    from_holes = 0
    for f in range(n):
        rb, rs = P.db_postprocess(maps[f], 2 * h, 2 * w)
        gb, gs = got[f]
        assert gb.shape == rb.shape and np.array_equal(gb, rb), (f, gb, rb)
        assert np.abs(gs - rs).max() < 1e-6 if len(rs) else True
        keep, P._hole_contours = P._hole_contours, lambda m: []
        try:
            from_holes += len(rb) - len(P.db_postprocess(maps[f], 2 * h, 2 * w)[0])
        finally:
            P._hole_contours = keep
    assert from_holes >= 3                  # the maps do exercise the hole path (3 / 8 / 7 boxes for the three seeds)


def test_prepost_fuzz_sample(ctx, capsys):
    """A fixed-seed sample of tools/fuzz_prepost.py: det pre-process at odd frame sizes and the perspective crop + resize on
    random quads (rotated, skewed, partly outside the frame, 1-3 pixel sides, tall, very wide), all bit-exact.  The tool found
    the rounding ties of integer-cornered quads (homography solved / evaluated in a different operation order); since then the
    engine and the oracle both follow cv2's LU + blocked evaluation, and 2900 crops / 290 frame sizes compare equal."""
    import importlib.util
    import os
    import sys
    spec = importlib.util.spec_from_file_location("fuzz_prepost", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_prepost.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    argv, sys.argv = sys.argv, ["fuzz_prepost", "--cases", "150", "--seed", "77"]
    try:
        rc = fz.main()
    finally:
        sys.argv = argv
    assert rc == 0, capsys.readouterr().out


def test_db_fuzz_sample(ctx, capsys):
    """A fixed-seed sample of tools/fuzz_db.py (rotated rectangles, ellipses, thin strokes, frame-touching blobs, dips, noise;
    random map / source sizes, thresholds and unclip ratios): identical integer boxes, scores within 1e-6.  The tool found that
    a float32 unclip_ratio in the C ABI moves `area * ratio / perimeter` off the reference's double value for ratios like 1.2
    (a .5 corner then rounds the other way); box_thresh and unclip_ratio are doubles in vse_db_params since."""
    import importlib.util
    import os
    import sys
    spec = importlib.util.spec_from_file_location("fuzz_db", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_db.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    argv, sys.argv = sys.argv, ["fuzz_db", "--cases", "120", "--seed", "31"]
    try:
        rc = fz.main()
    finally:
        sys.argv = argv
    assert rc == 0, capsys.readouterr().out


@pytest.mark.parametrize("mid,hilo", [("V3_ch_det_fast", True), ("V4_ch_det_fast", False), ("V4_ch_det", False), ("V2_ch_det", False)])
def test_stem_with_fused_preprocessing_is_bit_identical(ctx, mid, hilo):
    """compile_model(fuse_preprocess=True): the detector's stem conv resizes the uint8 frames itself (F_U8SRC, vse_det_forward
    without a pre-processing pass).  Same integer bilinear arithmetic, same fp16 inputs, same K order -> the probability maps
    equal those of the pre-processing pass + plan BIT FOR BIT: 1080p -> 544 x 960, 720p, a frame that needs no resize, and a
    row-pitched view (`frame[cropped:]`, backend/tools/subtitle_ocr.py:283)."""
    import torch
    from oracle import net_ref
    from vse_amd import engine, pipeline, synth
    desc, w = net_ref.get_weights(mid)
    nets = {f: engine.Net(ctx, desc, w, fetch_cols=(0,), hilo=hilo, input_norm=pipeline.DET_NORM, fuse_preprocess=f) for f in (False, True)}
    cases = [synth.make_frames(2, 1080, 1920, seed=4), synth.make_frames(1, 720, 1280, seed=5), synth.make_frames(3, 96, 160, seed=6)]
    for frames in cases:
        dev = torch.from_numpy(frames).cuda()
        views = [dev, dev[:, dev.shape[1] // 2:, :, :]]                       # whole frames; bottom half (non-contiguous frame stride)
        for v in views:
            rh, rw = pipeline.det_resize_shape(v.shape[1], v.shape[2])
            a = nets[False].det_forward(v, rh, rw)
            b = nets[True].det_forward(v, rh, rw)
            assert nets[True].fuse_preprocess, "the stem of this detector must take the fused pre-processing"
            assert torch.equal(a, b), (mid, tuple(v.shape))
    # the fused plan refuses a pre-processed tensor without a frame geometry
    fresh = engine.Net(ctx, desc, w, fetch_cols=(0,), hilo=hilo, input_norm=pipeline.DET_NORM, fuse_preprocess=True)
    x = ctx.det_preprocess(torch.from_numpy(cases[2]).cuda(), 96, 160, raw=True)
    with pytest.raises(engine.VseError, match="vse_plan_set_source"):
        fresh.run(x)
