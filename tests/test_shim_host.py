"""Host-side behaviour of the drop-in shim that needs no GPU: dictionaries, kwargs, error behaviour."""
import os

import numpy as np
import pytest

from vse_amd import shim


@pytest.fixture(autouse=True)
def _restore_config():
    keep = dict(vars(shim.config))
    yield
    for k, v in keep.items():
        setattr(shim.config, k, v)


def test_en_charset_is_builtin():
    cs = shim.charset_for("en", 97)
    assert len(cs) == 97 and cs[0] == "blank" and cs[1] == "0" and cs[-1] == " " and cs[-2] == " "
    assert "".join(cs[1:11]) == "0123456789" and cs[cs.index("~") + 1] == "!"


def test_no_dictionary_is_an_error_not_invented_text():
    """Real converted weights + no dictionary must not return plausible-looking wrong text."""
    shim.config.allow_standin_weights = False
    shim.config.dict_dir = None
    with pytest.raises(FileNotFoundError, match="ppocr_keys_v1.txt"):
        shim.charset_for("ch", 6625)
    with pytest.raises(FileNotFoundError, match="korean_dict.txt"):
        shim.charset_for("korean", 3690)
    with pytest.raises(FileNotFoundError, match="latin_dict.txt"):
        shim.charset_for("fr", 187)
    shim.config.allow_standin_weights = True
    assert len(shim.charset_for("ch", 6625)) == 6625          # placeholder table only on explicit request


def test_dictionary_from_path_and_dict_dir(tmp_path):
    chars = ["的", "一", "是", "A", "#"]
    p = tmp_path / "ppocr_keys_v1.txt"
    p.write_bytes("\n".join(chars).encode("utf-8") + b"\n")
    want = ["blank"] + chars + [" "]
    assert shim.charset_for("ch", 7, rec_char_dict_path=str(p)) == want
    assert shim.charset_for("ch", 6, rec_char_dict_path=str(p), use_space_char=False) == want[:-1]
    shim.config.dict_dir = str(tmp_path)
    assert shim.charset_for("ch", 7) == want
    with pytest.raises(ValueError, match="7 classes but the recogniser has 6625"):
        shim.charset_for("ch", 6625)
    # CRLF files and the dict/ sub-directory of paddleocr's layout
    os.makedirs(tmp_path / "dict")
    (tmp_path / "dict" / "korean_dict.txt").write_bytes("가\r\n나\r\n".encode("utf-8"))
    assert shim.charset_for("korean", 4) == ["blank", "가", "나", " "]


def test_paddleocr_kwargs_are_checked_before_any_model_is_loaded():
    with pytest.raises(NotImplementedError):
        shim.PaddleOCR(det_model_dir="V4_ch_det", rec_model_dir="V4_ch_rec", use_angle_cls=True)
    with pytest.raises(NotImplementedError):
        shim.PaddleOCR(det_model_dir="V4_ch_det", rec_model_dir="V4_ch_rec", det_algorithm="EAST")
    with pytest.raises(TypeError, match="rec_char_dict"):
        shim.PaddleOCR(det_model_dir="V4_ch_det", rec_model_dir="V4_ch_rec", rec_char_dict="typo.txt")
    # the reference's own kwargs (backend/tools/ocr.py:91-113) pass the check and reach model loading
    shim.config.allow_standin_weights = False
    with pytest.raises(FileNotFoundError, match="weights for V4_ch_det"):
        shim.PaddleOCR(use_gpu=True, gpu_mem=500, det_algorithm="DB", det_model_dir="V4_ch_det", rec_algorithm="CRNN",
                       rec_batch_num=6, rec_model_dir="V4_ch_rec", max_batch_size=10, det=True, use_angle_cls=False,
                       drop_score=0, lang="ch", ocr_version="PP-OCRv4", rec_image_shape="3,48,320", use_onnx=False,
                       onnx_providers=[])


def test_bucketed_groups_fold_small_buckets_upwards():
    """Bucketed recognition (throughput mode): a width bucket with fewer than min_rec_group crops absorbs the next narrower
    bucket — widest first, because only a wider group can hold narrower crops; every crop stays in exactly one group whose
    width covers it, and min_rec_group=0 keeps the plain buckets."""
    from vse_amd import pipeline
    p = pipeline.OcrPipeline.__new__(pipeline.OcrPipeline)
    p.rec_mode, p.rec_h, p.rec_base_w, p.bucket, p.max_rec_batch = "bucketed", 48, 320, 256, 64
    ratios = [10] * 28 + [15] * 28 + [20] * 22 + [26] * 4
    specs = [dict(ratio=r) for r in ratios]
    p.min_rec_group = 0
    assert [(len(i), w) for i, w, _ in p._groups(specs)] == [(28, 512), (28, 768), (22, 1024), (4, 1280)]
    p.min_rec_group = 8
    groups = p._groups(specs)
    assert [(len(i), w) for i, w, _ in groups] == [(28, 512), (28, 768), (26, 1280)]
    assert all(ws == [w] * len(idx) for idx, w, ws in groups)          # bucketed: every crop padded to the bucket's width
    seen = sorted(i for idx, _, _ in groups for i in idx)
    assert seen == list(range(len(specs)))
    assert all(48 * ratios[i] <= w for idx, w, _ in groups for i in idx)
    # cascades: every bucket too small -> one group at the widest width; more than max_rec_batch crops are chunked as before
    p.min_rec_group = 8
    assert [(len(i), w) for i, w, _ in p._groups([dict(ratio=r) for r in [10] * 3 + [15] * 2 + [20] * 2 + [26]])] == [(8, 1280)]
    p.max_rec_batch = 16
    assert [len(i) for i, _, _ in p._groups([dict(ratio=10)] * 40)] == [16, 16, 8]


def test_ragged_groups_keep_every_crop_at_its_reference_width():
    """Ragged recognition (default): a crop's width is the padded width of ITS reference chunk (per frame, sorted by w/h,
    chunks of rec_batch_num, chunk as wide as its widest member, at least 320: oracle.pipeline_ref.rec_batches ==
    paddleocr's TextRecognizer loop behind backend/tools/ocr.py:27), while crops of all frames share the launch groups."""
    from oracle import pipeline_ref as P
    from vse_amd import pipeline
    rng = np.random.default_rng(4)
    p = pipeline.OcrPipeline.__new__(pipeline.OcrPipeline)
    p.rec_h, p.rec_base_w, p.bucket, p.max_rec_batch, p.rec_batch_num, p.min_rec_group = 48, 320, 256, 64, 6, 8
    specs, want = [], {}
    for f in range(30):
        k = int(rng.integers(0, 9))
        sizes = [(int(rng.integers(20, 900)), int(rng.integers(16, 60))) for _ in range(k)]
        crops = [np.zeros((h, w, 3), np.uint8) for w, h in sizes]
        base = len(specs)
        for w, h in sizes:
            specs.append(dict(frame=f, ratio=w / float(h)))
        for idx, img_w in P.rec_batches(crops, 6):
            for i in idx:
                want[base + i] = int(img_w)
    p.rec_mode = "reference"
    ref_groups = p._groups(specs)
    assert all(len(idx) <= 6 and len({specs[i]["frame"] for i in idx}) == 1 and ws == [w] * len(idx) for idx, w, ws in ref_groups)
    assert {i: w for idx, w, _ in ref_groups for i in idx} == want
    p.rec_mode = "ragged"
    groups = p._groups(specs)
    assert sorted(i for idx, _, _ in groups for i in idx) == list(range(len(specs)))
    assert {i: wi for idx, _, ws in groups for i, wi in zip(idx, ws)} == want          # same width as in the reference
    assert all(max(ws) <= w < max(ws) + 64 and w % 64 == 0 for _, w, ws in groups)     # the tensor just covers its widest sample
    assert all(len(idx) <= 64 for idx, _, _ in groups)
    assert len(groups) <= 8 and len(ref_groups) >= 30                                  # ... in a handful of launches
