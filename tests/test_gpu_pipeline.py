"""End-to-end: frames -> boxes + strings through the drop-in call sites vs the oracle text system."""
import numpy as np
import pytest

from oracle import net_ref
from oracle import pipeline_ref as P

pytestmark = pytest.mark.gpu


def _iou(a, b):
    ax0, ay0, ax1, ay1 = a[:, 0].min(), a[:, 1].min(), a[:, 0].max(), a[:, 1].max()
    bx0, by0, bx1, by1 = b[:, 0].min(), b[:, 1].min(), b[:, 0].max(), b[:, 1].max()
    iw, ih = max(0, min(ax1, bx1) - max(ax0, bx0)), max(0, min(ay1, by1) - max(ay0, by0))
    u = (ax1 - ax0) * (ay1 - ay0) + (bx1 - bx0) * (by1 - by0) - iw * ih
    return iw * ih / u if u > 0 else 1.0


FLOOR_EN_FAST = 0.3     # share of identical strings, V4_en_rec_fast stand-in: measured 3/5, 2/5, 3/3 (frames) and 5/7 (call site) on MI355X, round 6


@pytest.mark.parametrize("hw", [(720, 1280), (1080, 1920), (2160, 3840)])      # BASELINE configs C1 / C2 / C3 frame sizes
def test_ocr_pipeline_vs_oracle(ctx, hw):
    """Boxes: IoU >= 0.99 (in fact identical integers).  Strings: identical after CTC collapse, except that a
    time step may take another class where the ORACLE's own log-margin over it is below tests/parity.py TOL tie (1e-1) — the recogniser weights are
    stand-ins (the reference's blobs are missing), so its softmax is nearly flat and fp16 noise can flip such ties."""
    import torch
    from parity import check_text
    from vse_amd import pipeline, synth
    det = net_ref.get_weights("V3_ch_det_fast")            # the one model with real weights
    rec = net_ref.get_weights("V4_en_rec_fast")            # calibrated stand-in weights
    charset = P.en_charset()
    frames = synth.make_frames(3 if hw[0] < 2000 else 2, hw[0], hw[1], seed=hw[0], p_two_lines=0.5)
    pipe = pipeline.OcrPipeline(ctx, det, rec, charset, rec_mode="reference")
    dev = torch.from_numpy(frames).cuda()
    got = pipe.ocr(dev)
    nbox = 0
    exact = 0
    for f in range(len(frames)):
        x, _ = P.det_preprocess(frames[f])
        prob = net_ref.run_graph(det[0], det[1], x)[0].numpy()[0, 0]
        rb, _ = P.db_postprocess(prob, hw[0], hw[1])
        rb = P.sorted_boxes(rb)
        gb, gr = got[f]
        assert len(gb) == len(rb)
        for a, b in zip(gb, rb):
            assert _iou(np.asarray(a), np.asarray(b)) >= 0.99          # north_star: box IoU >= 0.99
        crops = [P.get_rotate_crop_image(frames[f], b) for b in rb]
        for idx, img_w in P.rec_batches(crops, 6):
            batch = np.stack([P.resize_norm_img(crops[i], img_w) for i in idx])
            probs = net_ref.run_graph(rec[0], rec[1], batch)[0].numpy()
            for k, i in enumerate(idx):
                ids, conf = P.ctc_greedy(probs[k])
                ref_text = P.decode_text(ids, charset)
                text, score = gr[i]
                # the string must be reachable from the oracle's per-step distribution through near-ties only (tests/parity.py: any other
                # character fails); an identical string carries the oracle's confidence within TOL maxp_rel (12 %; measured <= 2 %), a different one within conf_diff (5 %)
                exact += check_text("V4_en_rec_fast", text, score, probs[k], charset, ref_text, conf)
                nbox += 1
    print(f"{hw}: {exact} / {nbox} strings identical, the rest reachable through near-ties of the oracle's distribution")
    # (identical strings on EVERY crop cannot be asked of a random-weight head: a step flips when the oracle's top-2 log-margin is under the
    # engine's error — ~0.3-0.5 % of the steps — and a 1000-px crop has 125 steps; reachability above is the criterion with teeth)
    from parity import CONF_DIFFS, share_floor
    print(f"{hw}: reachable-but-different strings so far: {len(CONF_DIFFS)}, largest relative confidence difference {max(CONF_DIFFS, default=0.0):.3g}")
    assert nbox >= 2
    share_floor(exact, nbox, FLOOR_EN_FAST, hw)


def test_blank_and_mixed_frames(ctx):
    """Frames without text give empty results (arrays with len 0, as the reference's callers expect), also in the middle
    of a batch; the frames with text are unaffected by their neighbours."""
    import torch
    from vse_amd import pipeline, synth
    det = net_ref.get_weights("V3_ch_det_fast")
    rec = net_ref.get_weights("V4_en_rec_fast")
    charset = P.en_charset()
    text = synth.make_frames(2, 720, 1280, seed=5, p_two_lines=1.0)
    blank = np.full((720, 1280, 3), 40, np.uint8)
    frames = np.stack([blank, text[0], blank, text[1], np.zeros_like(blank)])
    pipe = pipeline.OcrPipeline(ctx, det, rec, charset, rec_mode="reference")
    got = pipe.ocr(torch.from_numpy(frames).cuda())
    alone = pipe.ocr(torch.from_numpy(text).cuda())
    for f in (0, 2, 4):
        x, _ = P.det_preprocess(frames[f])
        prob = net_ref.run_graph(det[0], det[1], x)[0].numpy()[0, 0]
        assert len(P.db_postprocess(prob, 720, 1280)[0]) == 0          # the oracle finds nothing either
        assert len(got[f][0]) == 0 and len(got[f][1]) == 0
    for f, g in ((1, 0), (3, 1)):
        assert len(got[f][0]) == len(alone[g][0]) >= 1
        assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(got[f][0], alone[g][0]))
        assert [t for t, _s in got[f][1]] == [t for t, _s in alone[g][1]]


def test_drop_in_call_sites(ctx):
    """SubtitleDetect.detect_subtitle / OcrRecogniser.predict / get_coordinates keep the reference's contracts."""
    import torch
    from vse_amd import shim, synth
    shim.config.language, shim.config.mode, shim.config.allow_standin_weights = "en", "fast", True
    frames = synth.make_frames(1, 720, 1280, seed=2)
    sd = shim.SubtitleDetect.__new__(shim.SubtitleDetect)
    from types import SimpleNamespace
    sd.text_detector = shim.TextDetector(SimpleNamespace(det_model_dir="V3_ch_det_fast", det_algorithm="DB"))
    dt_boxes, elapse = sd.detect_subtitle(frames[0])
    assert isinstance(dt_boxes, np.ndarray) and dt_boxes.dtype == np.float32 and dt_boxes.shape[1:] == (4, 2)
    assert elapse > 0 and len(dt_boxes) >= 1
    coords = shim.get_coordinates(dt_boxes.tolist())
    assert all(len(c) == 4 and c[0] < c[1] and c[2] < c[3] for c in coords)
    assert shim.get_coordinates(dt_boxes) == []                    # ndarray input -> [] like the reference
    # sliced, non-owning view (subtitle_ocr.py:283)
    half, _ = sd.detect_subtitle(frames[0][360:])
    assert half.shape[1:] == (4, 2)
    ocr = shim.OcrRecogniser()
    ocr.recogniser = shim.PaddleOCR(det_model_dir="V3_ch_det_fast", rec_model_dir="V4_en_rec_fast", drop_score=0,
                                    lang="en")
    boxes, res = ocr.predict(frames[0])
    assert len(boxes) == len(res) >= 1
    for b, (text, score) in zip(boxes, res):
        assert len(b) == 4 and all(isinstance(v, int) for p in b for v in p)
        assert isinstance(text, str) and 0.0 <= score <= 1.0
    blank = np.zeros((360, 640, 3), np.uint8)
    e_boxes, e_res = ocr.predict(blank)
    assert len(e_boxes) == 0 and len(e_res) == 0
    e_dt, _ = sd.detect_subtitle(blank)
    assert e_dt.shape == (0, 4, 2) and e_dt.tolist() == []


def test_accurate_mode_selector_on_engine(ctx):
    """a13 end to end: batched detector + on-demand OCR drive the reference's start/end-of-subtitle automaton; the
    batched selector and the statement-by-statement oracle loop agree when fed by the same engine."""
    import torch
    from types import SimpleNamespace
    from oracle import frame_loop_ref
    from vse_amd import frame_select, shim, synth
    shim.config.language, shim.config.mode, shim.config.allow_standin_weights = "en", "fast", True
    h, w = 360, 640
    lit = synth.make_frames(3, h, w, seed=4)
    dark = np.full((h, w, 3), 40, np.uint8)
    clip = [dark, dark, lit[0], lit[0], lit[0], dark, lit[1], lit[1], lit[2], lit[2], dark]
    det = shim.TextDetector(SimpleNamespace(det_model_dir="V3_ch_det_fast", det_algorithm="DB"))
    ocr = shim.OcrRecogniser()
    ocr.recogniser = shim.PaddleOCR(det_model_dir="V3_ch_det_fast", rec_model_dir="V4_en_rec_fast", drop_score=0, lang="en")
    area = SimpleNamespace(ymin=int(0.7 * h), ymax=h, xmin=0, xmax=w)

    def detect_batch(frames):
        dev = torch.from_numpy(np.stack(frames)).cuda()
        return [np.asarray(b, np.float32).reshape(-1, 4, 2) for b in det.batch(dev)]
    sel = frame_select.AccurateFrameSelector(detect_batch, ocr.predict, area, len(clip), 80, chunk=4)
    tasks = sel.run(clip)
    ref = frame_loop_ref.extract_frame_by_det(clip, len(clip), lambda f: detect_batch([f])[0], ocr.predict,
                                              dict(ymin=area.ymin, ymax=area.ymax, xmin=area.xmin, xmax=area.xmax), 80)
    assert [(t[0], t[1], t[2] is None) for t in tasks] == [(t[0], t[1], t[2] is None) for t in ref]
    nos = [t[1] for t in tasks]
    assert 3 in nos and 5 in nos and nos == sorted(nos)        # first subtitle spans frames 3..5


def test_streaming_form_is_identical(ctx):
    """ocr_stream(): the detector of batch k+1 overlaps post-processing + recognition of batch k (second HIP stream,
    alternate workspace slot); every batch's result equals the one-batch-at-a-time result, odd batch counts included."""
    import torch
    from vse_amd import pipeline, synth
    det = net_ref.get_weights("V3_ch_det_fast")
    rec = net_ref.get_weights("V4_en_rec_fast")
    pipe = pipeline.OcrPipeline(ctx, det, rec, P.en_charset(), rec_mode="bucketed")
    batches = [torch.from_numpy(synth.make_frames(3, 720, 1280, seed=40 + k, p_two_lines=0.5)).cuda() for k in range(5)]
    # a clip may change shape between batches (a short last batch, another resolution): plans and workspace slots are per shape
    batches[2] = torch.from_numpy(synth.make_frames(2, 540, 960, seed=47, p_two_lines=0.5)).cuda()
    batches[4] = batches[4][:1]
    seq = [pipe.ocr(b) for b in batches]
    for depth in (1, 2, 3):                         # detector batches in flight (default 2): slots and streams rotate
        got = list(pipe.ocr_stream(iter(batches), depth=depth))
        assert len(got) == 5
        for a, b in zip(seq, got):
            assert len(a) == len(b)
            for (ab, ar), (bb, br) in zip(a, b):
                assert len(ab) == len(bb) and all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(ab, bb))
                assert ar == br
    assert list(pipe.ocr_stream(iter([]))) == []


def test_box_parity_rate_real_detector(ctx):
    """Many frames instead of three (tools/parity_sweep.py at test size; the 18 frames include the three whose boxes sat a pixel row
    off in rounds 1-3): with the mobile detector's default precision — fp16 hi + lo weights, 1x1 / depthwise chains in LDS, hi + lo
    pair tensors between ops (OcrPipeline(det_weights="auto", det_chains=None)) — EVERY box is the oracle's integers: north_star's
    IoU >= 0.99 holds on all of them (round 4; the 128-frame sweeps at 720p and 1080p: 383 / 383 identical).  The
    layer-by-layer program (det_chains=False) keeps the looser bound it had."""
    import torch
    from vse_amd import pipeline, synth
    det = net_ref.get_weights("V3_ch_det_fast")
    rec = net_ref.get_weights("V4_en_rec_fast")
    frames = synth.make_frames(96, 1080, 1920, seed=777, p_two_lines=0.5)
    frames = np.concatenate([frames[16:34], frames[78:90]])      # includes frames 31, 80 and 87 of the sweep (the three hard ones)
    refs = []
    for f in range(len(frames)):
        x, _ = P.det_preprocess(frames[f])
        prob = net_ref.run_graph(det[0], det[1], x)[0].numpy()[0, 0]
        refs.append(P.sorted_boxes(P.db_postprocess(prob, 1080, 1920)[0]))
    for chains in (None, False):
        pipe = pipeline.OcrPipeline(ctx, det, rec, P.en_charset(), rec_mode="reference", det_chains=chains)
        assert pipe.det_weights == "fp16x2"
        boxes = pipe.detect(torch.from_numpy(frames).cuda())
        n = same = low = 0
        for f in range(len(frames)):
            gb = pipeline.sorted_boxes(boxes[f])
            assert len(gb) == len(refs[f])
            for a, b in zip(gb, refs[f]):
                n += 1
                same += int(np.array_equal(np.asarray(a), np.asarray(b)))
                low += int(_iou(np.asarray(a), np.asarray(b)) < 0.99)
        print(f"det_chains={chains}: {n} boxes, {same} identical, {low} below IoU 0.99")
        if chains is None:
            assert n >= 30 and same == n and low == 0, (n, same, low)          # north_star: IoU >= 0.99 on every box
        else:
            # fp16 tensors between the layers: a few boxes sit a pixel row off (3 - 4 of 43 measured, which ones moves with every
            # change of a rounding point, e.g. the depthwise conv fused in front of its 1x1 consumer)
            assert same >= 0.85 * n and low <= 0.12 * n, (n, same, low)


@pytest.mark.parametrize("mode", ["bucketed", "reference", "ragged"])
def test_recognizer_side_streams_keep_results(ctx, mode):
    """rec_streams > 1: width groups run on side streams.  Groups that share one plan key (n, h, w) — a bucket split into
    max_rec_batch chunks, reference-mode chunks of equal shape — are in flight at the same time and must each own a
    workspace slot: results equal the single-stream run."""
    import torch
    from vse_amd import pipeline, synth
    det = net_ref.get_weights("V3_ch_det_fast")
    rec = net_ref.get_weights("V4_en_rec_fast")
    frames, truth = synth.make_frames(6, 720, 1280, seed=11, p_two_lines=1.0, return_truth=True)
    quads = [[np.array([[x0 - 4, y0 - 4], [x1 + 4, y0 - 4], [x1 + 4, y1 + 4], [x0 - 4, y1 + 4]], np.float32)
              for (x0, y0, x1, y1, _t) in tr] for tr in truth]
    # every box three times: 36 crops, buckets of >= 8 crops split into chunks of 4 -> several groups per plan key
    quads = [q + q + q for q in quads]
    dev = torch.from_numpy(frames).cuda()
    pipe = pipeline.OcrPipeline(ctx, det, rec, P.en_charset(), rec_mode=mode, bucket=256, max_rec_batch=4, rec_batch_num=2)
    groups = pipe._groups(pipe._crop_specs(quads))
    keys = [(len(idx), w) for idx, w, _ in groups]
    assert len(keys) > len(set(keys)), "the case must contain groups that share a plan key"
    pipe.rec_streams = 1
    want = pipe.recognize(dev, quads)
    for streams in (2, 3):
        pipe.rec_streams = streams
        for _ in range(3):
            assert pipe.recognize(dev, quads) == want
    assert torch.cuda.current_stream(ctx.tdev) == torch.cuda.default_stream(ctx.tdev)


def test_text_recognizer_call_site(ctx, tmp_path, monkeypatch):
    """paddleocr TextRecognizer(args)(img_list): ready-made crops of different sizes, grouped like the reference (sorted
    by w/h, chunks of rec_batch_num, padded to the chunk's widest) — checked against the oracle recogniser."""
    from types import SimpleNamespace
    from vse_amd import shim, synth
    # the call site loads its weights by model id: hand it the oracle's CALIBRATED stand-ins as an .npz under config.weights_dir (the
    # seeded stand-ins the shim would otherwise fall back to — modelzoo.random_weights, "values do not matter" — answer every input alike)
    np.savez(tmp_path / "V4_en_rec_fast.npz", **net_ref.get_weights("V4_en_rec_fast")[1])
    monkeypatch.setattr(shim.config, "weights_dir", str(tmp_path), raising=False)          # restored by the fixture whatever happens below
    monkeypatch.setattr(shim.config, "allow_standin_weights", False, raising=False)
    frames, truth = synth.make_frames(3, 720, 1280, seed=21, p_two_lines=1.0, return_truth=True)
    crops = []
    for f, tr in enumerate(truth):
        for (x0, y0, x1, y1, _t) in tr:
            crops.append(np.ascontiguousarray(frames[f][y0 - 3:y1 + 3, x0 - 3:x1 + 3]))
    crops.append(np.ascontiguousarray(crops[0][:, :40]))          # a short one: padded to the 320-px base width
    tr = shim.TextRecognizer(SimpleNamespace(rec_model_dir="V4_en_rec_fast", rec_image_shape="3,48,320", lang="en",
                                             rec_batch_num=3))
    got, elapse = tr(crops)
    assert len(got) == len(crops) and elapse > 0
    # the identity-quad route is a pixel copy: the recogniser input equals the oracle's resize of the raw crop bit for bit
    import torch
    mh, mw = max(c.shape[0] for c in crops), max(c.shape[1] for c in crops)
    canvas = np.zeros((len(crops), mh, mw, 3), np.uint8)
    specs = []
    for i, c in enumerate(crops):
        h, w = c.shape[:2]
        canvas[i, :h, :w] = c
        specs.append(dict(quad=np.array([[0, 0], [w, 0], [w, h], [0, h]], np.float32), frame=i, crop_w=w, crop_h=h,
                          resized_w=P.rec_resized_width(w, h, 640), rotate=0))
    pre = ctx.rec_preprocess(torch.from_numpy(canvas).cuda(), specs, 48, 640).cpu().numpy()
    for i, c in enumerate(crops):
        ref = P.resize_norm_img(c, 640).transpose(1, 2, 0).astype(np.float16)
        assert np.array_equal(pre[i, ..., :3], ref), i
    from parity import check_text
    # the oracle runs the weights THE CALL SITE loaded (until round 5 this test compared the shim's uncalibrated stand-ins with the
    # oracle's calibrated ones under a criterion loose enough not to notice)
    rec = shim._load_model("V4_en_rec_fast")
    assert all(np.array_equal(rec[1][k], v) for k, v in net_ref.get_weights("V4_en_rec_fast")[1].items())
    charset = P.en_charset()
    exact = 0
    for idx, img_w in P.rec_batches(crops, 3):
        batch = np.stack([P.resize_norm_img(crops[i], img_w) for i in idx])
        probs = net_ref.run_graph(rec[0], rec[1], batch)[0].numpy()
        for k, i in enumerate(idx):
            ids, conf = P.ctc_greedy(probs[k])
            text, score = got[i]
            exact += check_text("V4_en_rec_fast", text, score, probs[k], charset, P.decode_text(ids, charset), conf)
    from parity import share_floor
    print(f"TextRecognizer call site: {exact} / {len(crops)} strings identical")
    share_floor(exact, len(crops), FLOOR_EN_FAST, "TextRecognizer call site")
    # (the stand-in recogniser's softmax is nearly flat, so an exactly equal string is not guaranteed on a handful of crops:
    # what this call site adds over the net-level parity tests — the crop route above and the grouping — is checked exactly)
    want_groups = [(list(idx), int(w)) for idx, w in P.rec_batches(crops, 3)]
    gspecs = [dict(frame=i, gframe=0, ratio=c.shape[1] / float(c.shape[0])) for i, c in enumerate(crops)]
    assert [(list(idx), int(w)) for idx, w, _ in tr.pipe._reference_chunks(gspecs) and
            [(c, w, None) for c, w in tr.pipe._reference_chunks(gspecs)]] == want_groups
    # the default (ragged) grouping gives every crop exactly that chunk width, and the reference launch structure the same answers
    assert {i: wi for idx, _, ws in tr.pipe._groups(gspecs) for i, wi in zip(idx, ws)} == {i: int(w) for idx, w in want_groups for i in idx}
    tr.pipe.rec_mode = "reference"
    assert tr(crops)[0] == got
    assert tr([])[0] == []


def test_extractor_on_a_clip_engine_vs_oracle(ctx):
    """BASELINE configs[0] as far as it can exist here: a short clip -> frame selection -> batched det + rec on the engine ->
    raw.txt filters -> SRT (vse_amd.extractor).
      * fps sampler, no area: which frames are looked at and which boxes survive the scene-text filter depend on geometry only,
        so the engine-fed run gives the CPU oracle's (frame number, coordinates) sequence exactly (boxes are identical integers);
      * strings of a stand-in recogniser are near-ties between two classes (its own output flips between crops), so string
        equality against the oracle is what the net-level parity tests bound, not this one; here the BATCHED engine run must
        equal the engine run frame by frame (the reference's order of work) byte for byte — raw.txt and SRT, in the fps mode
        and in the accurate mode (detector-driven selection, cached OCR results)."""
    import torch
    from vse_amd import extractor, pipeline, shim, staging, synth
    det = net_ref.get_weights("V3_ch_det_fast")
    rec = net_ref.get_weights("V4_en_rec_fast")
    cs = P.en_charset()
    pipe = pipeline.OcrPipeline(ctx, det, rec, cs, rec_mode="reference")

    class EngineOcr:
        def predict(self, frame):
            b, r = pipe.ocr(torch.from_numpy(np.ascontiguousarray(frame)).cuda()[None])[0]
            return shim.OcrRecogniser.arrange(b, r)

    class EngineOcrBatched(EngineOcr):
        def predict_batch(self, frames):
            return [shim.OcrRecogniser.arrange(b, r) for b, r in pipe.ocr(frames)]

    class EngineOcrStreamed(EngineOcrBatched):
        def predict_stream(self, batches):
            for out in pipe.ocr_stream(batches):
                yield [shim.OcrRecogniser.arrange(b, r) for b, r in out]

        def predict_with_dets(self, frames, dets):
            return [shim.OcrRecogniser.arrange(b, r) for b, r in pipe.ocr_from_det(frames, dets)]

    def engine_detect_stream(batches):
        for dets in pipe.detect_stream(batches):
            yield [np.asarray(b, np.float32).reshape(-1, 4, 2) for b in dets]

    det_fn = lambda x: net_ref.run_graph(det[0], det[1], x)[0].numpy()[0, 0]
    rec_fn = lambda b: net_ref.run_graph(rec[0], rec[1], b)[0].numpy()

    class OracleOcr:
        def predict(self, frame):
            return P.ocr_predict_glue(*P.text_system(frame, det_fn, rec_fn, cs))

    def engine_detect(frames):
        dev = frames if torch.is_tensor(frames) else torch.from_numpy(np.stack(frames)).cuda()
        return [np.asarray(b, np.float32).reshape(-1, 4, 2) for b in pipe.detect(dev)]

    h, wd = 360, 640
    lit = synth.make_frames(3, h, wd, seed=4)
    dark = np.full((h, wd, 3), 40, np.uint8)
    clip = [dark, dark] + [lit[0]] * 5 + [dark] + [lit[1]] * 5 + [lit[2]] * 4 + [dark]
    src = extractor.ArraySource(clip, 12.0)
    area = extractor.SubtitleArea(ymin=int(0.7 * h), ymax=h, xmin=0, xmax=wd)
    geo = lambda lines: [ln.split("\t")[:2] for ln in lines]
    fps_kw = dict(sub_area=None, mode="fast", extract_frequency=6, drop_score=0.0, batch=8)
    ora = extractor.SubtitleExtractor(src, OracleOcr(), **fps_kw)
    ora.run()
    for kw in (fps_kw, dict(sub_area=area, mode="accurate", drop_score=0.0, batch=8)):
        one = extractor.SubtitleExtractor(src, EngineOcr(), detect_batch=lambda fr: sum((engine_detect([f]) for f in fr), []), **kw)
        many = extractor.SubtitleExtractor(src, EngineOcrBatched(), detect_batch=engine_detect, **kw)
        a, b = one.run(), many.run()
        assert many.raw_lines == one.raw_lines and len(one.raw_lines) >= 3
        assert a == b and a.count(" --> ") >= 1
        # the same through pinned staging + producer thread (device tensors reach detect_batch / predict_batch)
        staged = extractor.SubtitleExtractor(src, EngineOcrBatched(), detect_batch=engine_detect,
                                             uploader=staging.Uploader(ctx.tdev), **kw)
        assert staged.run() == a and staged.raw_lines == one.raw_lines
        # ... and with the detectors of the next batches in flight (predict_stream / detect_stream) and, in the accurate mode,
        # the wanted frames recognised from the boxes the selector's detector already produced
        streamed = extractor.SubtitleExtractor(src, EngineOcrStreamed(), detect_batch=engine_detect, detect_stream=engine_detect_stream,
                                               uploader=staging.Uploader(ctx.tdev), **kw)
        assert streamed.run() == a and streamed.raw_lines == one.raw_lines
        if kw is fps_kw:
            assert geo(many.raw_lines) == geo(ora.raw_lines)


def test_ragged_recognition_is_bit_identical_to_the_reference_grouping(ctx):
    """crops of many frames in a handful of launches (rec_mode="ragged") vs one launch sequence per <= 6-crop
    chunk of one frame (rec_mode="reference", backend/tools/ocr.py:99 + backend/config.py:58): every (text, score) pair is
    EQUAL — the score is the mean of the kept time steps' max probabilities, so equality means the same arg-max indices and
    the same probability bits — on 240 crops of mixed widths, 1 to 8 per frame, on the server and the mobile recogniser,
    single stream and side streams, several bucket / batch-rounding settings."""
    import torch
    from vse_amd import pipeline, synth
    det = net_ref.get_weights("V3_ch_det_fast")
    frames, truth = synth.make_frames(48, 720, 1280, seed=21, p_two_lines=0.5, return_truth=True)
    rng = np.random.default_rng(8)
    quads = []
    for tr in truth:
        qs = []
        for (x0, y0, x1, y1, _t) in tr:
            qs.append(np.array([[x0 - 4, y0 - 4], [x1 + 4, y0 - 4], [x1 + 4, y1 + 4], [x0 - 4, y1 + 4]], np.float32))
        while len(qs) < 8 and rng.random() < 0.85:          # sub-boxes of random widths: word-sized to line-sized crops
            x0, y0, x1, y1, _t = tr[int(rng.integers(len(tr)))]
            a = int(rng.integers(x0, max(x0 + 1, x1 - 40)))
            b = int(rng.integers(min(a + 30, x1), x1 + 1))
            qs.append(np.array([[a, y0 - 3], [b, y0 - 3], [b, y1 + 3], [a, y1 + 3]], np.float32))
        quads.append(qs)
    assert sum(len(q) for q in quads) >= 240
    dev = torch.from_numpy(frames).cuda()
    for rec_id, charset in (("V4_ch_rec", None), ("V4_en_rec_fast", P.en_charset())):
        rec = net_ref.get_weights(rec_id)
        from vse_amd import shim
        cs = charset or P.standin_charset(shim._ncls(rec[0]))
        pipe = pipeline.OcrPipeline(ctx, det, rec, cs, rec_mode="reference")
        want = pipe.recognize(dev, quads)
        assert sum(1 for r in want for (t, s) in r if t) > 100           # the comparison is not about empty strings
        pipe.rec_mode = "ragged"
        for bucket, rnd, mg, streams in ((256, 4, 8, 1), (64, 1, 0, 1), (128, 8, 4, 3), (1024, 1, 0, 2)):
            pipe.bucket, pipe.batch_round, pipe.min_rec_group, pipe.rec_streams = bucket, rnd, mg, streams
            got = pipe.recognize(dev, quads)
            assert got == want, (rec_id, bucket, rnd, mg, streams)
        pipe.rec_streams = 1
        # the bucket padding of rounds 1-2 (crops padded to the width of their bucket, NOT the reference's padding) on the same crops:
        # how many (text, score) pairs it changes — the reason it is no longer the benchmarked mode (VERDICT r2 #1c)
        pipe.rec_mode, pipe.bucket, pipe.batch_round, pipe.min_rec_group = "bucketed", 256, 4, 8
        other = pipe.recognize(dev, quads)
        flat_w = [r for fr in want for r in fr]
        flat_o = [r for fr in other for r in fr]
        n_text = sum(a[0] != b[0] for a, b in zip(flat_w, flat_o))
        n_any = sum(a != b for a, b in zip(flat_w, flat_o))
        print(f"{rec_id}: bucketed vs reference on {len(flat_w)} crops: {n_text} strings differ, {n_any} (text, score) pairs differ")
        assert n_any > 0, "bucket padding changes the padded width of most crops: some outputs must move"
        assert max(abs(a[1] - b[1]) for a, b in zip(flat_w, flat_o)) < 0.5
        pipe.rec_mode = "ragged"


def test_crops_of_several_batches_recognised_together(ctx):
    """recognize_multi / ocr_stream(rec_span=2): the crops of consecutive frame batches share the recogniser's launch
    sequences (ragged mode) — every (text, score) equals the per-batch run, and the streamed results keep their order."""
    import torch
    from vse_amd import pipeline, synth
    det = net_ref.get_weights("V3_ch_det_fast")
    rec = net_ref.get_weights("V4_en_rec_fast")
    pipe = pipeline.OcrPipeline(ctx, det, rec, P.en_charset(), bucket=256, batch_round=4)
    batches = [torch.from_numpy(synth.make_frames(5, 720, 1280, seed=60 + k, p_two_lines=0.6)).cuda() for k in range(5)]
    want = [pipe.ocr(b) for b in batches]
    assert sum(len(r[1]) for w in want for r in w) >= 25
    for span in (2, 3):
        got = list(pipe.ocr_stream(iter(batches), depth=2, rec_span=span))
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert len(g) == len(w)
            for (gb, gr), (wb, wr) in zip(g, w):
                assert gr == wr and all(np.array_equal(a, b) for a, b in zip(gb, wb))
    boxes = [[b for b, _ in w] for w in want]
    multi = pipe.recognize_multi(batches[:3], boxes[:3])
    assert multi == [[r for _, r in w] for w in want[:3]]


def test_recogniser_as_hip_graph_gives_the_same_results(ctx):
    """OcrPipeline.rec_graphs: every recogniser invocation (plan + CTC collapse) captured as one HIP graph against fixed buffers
    (vse_rec_graph_create / vse_graph_launch) — same (text, score) pairs as plain launches, run after run, ragged widths included."""
    import torch
    from vse_amd import pipeline, synth
    det = net_ref.get_weights("V3_ch_det_fast")
    rec = net_ref.get_weights("V4_en_rec_fast")
    pipe = pipeline.OcrPipeline(ctx, det, rec, P.en_charset(), bucket=256, batch_round=4)
    pipe.rec_streams = 2
    pipe.ragged_floor, pipe.ragged_launch_cost = 0, 300          # several width groups per call: they run on the side streams
    batches = [torch.from_numpy(synth.make_frames(8, 720, 1280, seed=70 + k, p_two_lines=0.6)).cuda() for k in range(3)]
    want = [pipe.ocr(b) for b in batches]
    pipe.rec_graphs = True
    for _ in range(2):
        got = [pipe.ocr(b) for b in batches]
        assert [[r for _, r in g] for g in got] == [[r for _, r in w] for w in want]
    assert getattr(pipe.rec, "_graphs", {}) and all(v["graph"] is not None for v in pipe.rec._graphs.values())
