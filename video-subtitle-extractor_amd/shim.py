"""Drop-in replacements for the reference's OCR call sites, backed by the HIP engine.

Mirrors (same names, argument meaning, return shapes, error behaviour):
  SubtitleDetect().detect_subtitle(img) -> (ndarray[N,4,2] float32, elapse)     backend/tools/subtitle_detect.py:5-26
  OcrRecogniser().predict(img) -> (list of 4 (x,y) int tuples, list of (text, score))   backend/tools/ocr.py:9-113
  get_coordinates(dt_box) -> list of (xmin,xmax,ymin,ymax)                       backend/tools/ocr.py:115-134
  PaddleModelConfig(hardware_accelerator)                                        backend/tools/paddle_model_config.py:7-106
  HardwareAccelerator                                                            backend/tools/hardware_accelerator.py:4-93
  TextDetector(args)(img), TextRecognizer(args)(img_list), PaddleOCR(**kw)(img, cls=False)   (paddleocr 2.10 API)
plus batch variants (`detect_subtitle_batch`, `predict_batch`) that the reference lacks (it never batches, F4).

`config` is a plain namespace with the knobs the reference reads from qfluentwidgets (backend/config.py:52-90).
"""
import os
import time
from types import SimpleNamespace

import numpy as np

from . import engine, modelzoo, paddle_io, pipeline

config = SimpleNamespace(language="ch", mode="fast", recBatchNumber=6, maxBatchSize=10, dropScore=75,
                         subtitleAreaDeviationRate=0, hardwareAcceleration=True, device=0,
                         allow_standin_weights=False, weights_dir=None, dict_dir=None, models_root=None)

LATIN_LANG = ['af', 'az', 'bs', 'cs', 'cy', 'da', 'de', 'es', 'et', 'fr', 'ga', 'hr', 'hu', 'id', 'is', 'it', 'ku',
              'la', 'lt', 'lv', 'mi', 'ms', 'mt', 'nl', 'no', 'oc', 'pi', 'pl', 'pt', 'ro', 'rs_latin', 'sk', 'sl',
              'sq', 'sv', 'sw', 'tl', 'tr', 'uz', 'vi', 'latin', 'german', 'french']
ARABIC_LANG = ['ar', 'fa', 'ug', 'ur']
CYRILLIC_LANG = ['ru', 'rs_cyrillic', 'be', 'bg', 'uk', 'mn', 'abq', 'ady', 'kbd', 'ava', 'dar', 'inh', 'che', 'lbe',
                 'lez', 'tab', 'cyrillic']
DEVANAGARI_LANG = ['hi', 'mr', 'ne', 'bh', 'mai', 'ang', 'bho', 'mah', 'sck', 'new', 'gom', 'sa', 'bgc', 'devanagari']
OTHER_LANG = ['ch', 'japan', 'korean', 'en', 'ta', 'kn', 'te', 'ka', 'chinese_cht']


class HardwareAccelerator:
    """backend/tools/hardware_accelerator.py: here the only accelerator is a gfx950 HIP device."""
    _instance = None

    @classmethod
    def instance(cls):
        if cls._instance is None:
            cls._instance = HardwareAccelerator()
            cls._instance.initialize()
        return cls._instance

    def __init__(self):
        self._hip = False
        self._enabled = True

    def initialize(self):
        try:
            import torch
            self._hip = bool(torch.cuda.is_available())
        except Exception:
            self._hip = False

    def has_accelerator(self):
        return self._enabled and self._hip

    def has_cuda(self):          # name kept for call-compatibility (backend/tools/ocr.py:92)
        return self.has_accelerator()

    @property
    def onnx_providers(self):
        return []

    @property
    def accelerator_name(self):
        return "MI355X (HIP)" if self.has_accelerator() else "CPU"

    def set_enabled(self, enable):
        self._enabled = enable


class PaddleModelConfig:
    """Model choice matrix of backend/tools/paddle_model_config.py:53-91 over the model ids under models/."""

    def __init__(self, hardware_accelerator):
        self.hardware_accelerator = hardware_accelerator
        lang = config.language
        self.REC_CHAR_TYPE = lang
        self.MODEL_VERSION = 'V4'
        self.REC_IMAGE_SHAPE = '3,48,320'
        ver = 'V4'
        det, rec = f'{lang}_det', f'{lang}_rec'
        if lang in LATIN_LANG + ARABIC_LANG + CYRILLIC_LANG + DEVANAGARI_LANG + OTHER_LANG:
            if config.mode == 'fast':
                det, rec = 'ch_det_fast', f'{lang}_rec_fast'
            elif config.mode == 'auto':
                if hardware_accelerator.has_accelerator():
                    det = 'ch_det'
                    rec = 'ch_rec' if lang == 'en' else f'{lang}_rec'
                else:
                    det, rec = 'ch_det_fast', f'{lang}_rec_fast'
            else:
                det, rec = 'ch_det', f'{lang}_rec'
            if not self._exists(ver, rec):
                rec = f'{lang}_rec_fast'
            if not self._exists(ver, rec):
                ver, rec = 'V3', f'{lang}_rec'
            if not self._exists(ver, rec):
                ver, rec = 'V3', f'{lang}_rec_fast'
            if lang in LATIN_LANG:
                rec = 'latin_rec_fast'
            elif lang in ARABIC_LANG:
                rec = 'arabic_rec_fast'
            elif lang in CYRILLIC_LANG:
                rec = 'cyrillic_rec_fast'
            elif lang in DEVANAGARI_LANG:
                rec = 'devanagari_rec_fast'
            self.MODEL_VERSION = ver
            self.REC_IMAGE_SHAPE = '3,32,320' if ver == 'V2' else '3,48,320'
            # the reference lists the chosen model directories here (paddle_model_config.py:100-106) and therefore
            # raises FileNotFoundError for a language without any model (e.g. 'kn'); keep that error behaviour
            for v, name in ((ver, rec), ('V4', det)):
                if not self._exists(v, name):
                    raise FileNotFoundError(f"no model {v}/{name} for language {lang!r}")
        self.DET_MODEL_PATH = f'V4_{det}'        # det path is computed while MODEL_VERSION == 'V4' (App. D)
        self.REC_MODEL_PATH = f'{ver}_{rec}'

    @staticmethod
    def _exists(ver, name):
        # the reference tests for the model DIRECTORY (paddle_model_config.py:66-79); config.models_root is that tree
        if config.models_root and os.path.exists(os.path.join(config.models_root, ver, name, "inference.pdmodel")):
            return True
        return os.path.exists(os.path.join(modelzoo.MODELS_DIR, f'{ver}_{name}.json'))

    def convertToOnnxModelIfNeeded(self, model_dir, *a, **k):   # identity: no ONNX path here
        return model_dir


# dictionary file of each recogniser language inside paddleocr 2.10 (ppocr/utils/...), relative to config.dict_dir
_DICT_FILES = {"ch": "ppocr_keys_v1.txt", "en": "en_dict.txt"}


def _dict_candidates(lang):
    group = ("latin" if lang in LATIN_LANG else "arabic" if lang in ARABIC_LANG else "cyrillic" if lang in CYRILLIC_LANG
             else "devanagari" if lang in DEVANAGARI_LANG else lang)
    name = _DICT_FILES.get(group, f"{group}_dict.txt")
    return [name, os.path.join("dict", name)]


def read_dict_file(path):
    """paddleocr BaseRecLabelDecode: one character per line (trailing newline / CR stripped, nothing else)."""
    with open(path, "rb") as f:
        return [ln.decode("utf-8").strip("\n").strip("\r\n") for ln in f.readlines()]


def standin_charset(lang, ncls):
    """Index-faithful placeholder table: ONLY for stand-in weights / index-level parity runs, never for real text."""
    if lang == "en" and ncls == 97:
        return en_charset()
    return ["blank"] + [chr(0x4E00 + i) for i in range(ncls - 2)] + [" "]


def en_charset():
    """paddleocr's 95-entry en_dict.txt order: digits .. '~', then '!' .. '/', then ' ' — plus CTC
    blank in front and the use_space_char blank behind = the 97 classes of V4/en_rec_fast."""
    chars = [chr(c) for c in range(0x30, 0x7F)] + [chr(c) for c in range(0x21, 0x30)] + [" "]
    return ["blank"] + chars + [" "]


def charset_for(lang, ncls, rec_char_dict_path=None, use_space_char=True):
    """CTC label table ['blank'] + dictionary (+ [' '] when use_space_char) for a recogniser with `ncls` classes
    (paddleocr CTCLabelDecode).  Resolution order: an explicit rec_char_dict_path (the PaddleOCR kwarg),
    the language's file under config.dict_dir, the built-in `en` table.  The dictionaries live inside the paddleocr wheel,
    not in the reference checkout: without one there is NO way to turn class indices into the right characters, so this
    raises instead of inventing text — unless config.allow_standin_weights asks for the placeholder table (index-level
    parity runs with stand-in weights)."""
    path = rec_char_dict_path
    if path is None and getattr(config, "dict_dir", None):
        for cand in _dict_candidates(lang):
            if os.path.exists(os.path.join(config.dict_dir, cand)):
                path = os.path.join(config.dict_dir, cand)
                break
    if path is not None:
        chars = ["blank"] + read_dict_file(path) + ([" "] if use_space_char else [])
        if len(chars) != ncls:
            raise ValueError(f"dictionary {path} gives {len(chars)} classes but the recogniser has {ncls}")
        return chars
    if lang == "en" and ncls == 97 and use_space_char:
        return en_charset()
    if config.allow_standin_weights:
        return standin_charset(lang, ncls)
    raise FileNotFoundError(f"no character dictionary for language {lang!r}: pass rec_char_dict_path=... or set "
                            f"config.dict_dir to a directory holding {_dict_candidates(lang)[0]} (paddleocr ppocr/utils)")


def _paddle_dir(model):
    """A Paddle inference model directory for `model`: the argument itself when it is one (det_model_dir / rec_model_dir as the
    reference passes them, ocr.py:93-99), else <config.models_root>/<version>/<name> for an id like 'V4_ch_det' (the layout
    of the reference's backend/models)."""
    if os.path.isdir(model) and os.path.exists(os.path.join(model, "inference.pdmodel")):
        return model
    if config.models_root and "_" in os.path.basename(model):
        ver, name = os.path.basename(model).split("_", 1)
        d = os.path.join(config.models_root, ver, name)
        if os.path.exists(os.path.join(d, "inference.pdmodel")):
            return d
    return None


def _load_model(model):
    """-> (descriptor, weights).  `model` is a model id under models/ ('V4_ch_det') or a Paddle inference model directory.
    Weights come, in this order, from inference.pdiparams of the Paddle directory (read directly, paddle_io), from
    <id>.npz under config.weights_dir or models/, or — only when config.allow_standin_weights — from the seeded stand-ins."""
    pdir = _paddle_dir(model)
    if pdir is not None:
        mid = os.path.basename(model) if pdir != model else "_".join(os.path.normpath(pdir).split(os.sep)[-2:])
        desc, weights = paddle_io.load_model_dir(pdir, mid)
        if weights is not None:
            return desc, weights
        model_id = mid
    else:
        model_id = os.path.basename(model)
        desc = modelzoo.load_descriptor(model_id)
    cands = [os.path.join(d, model_id + ".npz") for d in (config.weights_dir, modelzoo.MODELS_DIR) if d]
    for p in cands:
        if os.path.exists(p):
            return desc, modelzoo.load_weights_npz(p)
    if config.allow_standin_weights:
        return desc, modelzoo.random_weights(desc)
    raise FileNotFoundError(f"weights for {model_id} not found (no inference.pdiparams in {pdir or config.models_root}, none of "
                            f"{cands}); the reference checkout ships them only for V3_ch_det_fast — point config.models_root at "
                            "a backend/models tree that holds the .pdiparams blobs, or pass the model directory itself")


def _ncls(desc):
    last = [op for op in desc["ops"] if op["type"] in ("matmul_v2", "matmul")][-1]
    return desc["params"][last["in"]["Y"][0]]["dims"][1]


_ctx = None


def _context():
    global _ctx
    if _ctx is None:
        _ctx = engine.Context(config.device)
    return _ctx


def _to_device(img):
    import torch
    if isinstance(img, torch.Tensor):
        return img if img.dim() == 4 else img[None]
    a = np.ascontiguousarray(img)          # accepts sliced views like frame[cropped:] (subtitle_ocr.py:283)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected uint8 BGR HxWx3 image")
    return torch.from_numpy(a).to(_context().tdev)[None]


class SubtitleDetect:
    """backend/tools/subtitle_detect.py:5-26."""

    def __init__(self):
        hw = HardwareAccelerator.instance()
        mc = PaddleModelConfig(hw)
        self.text_detector = TextDetector(SimpleNamespace(det_model_dir=mc.DET_MODEL_PATH, det_algorithm='DB'))

    def detect_subtitle(self, img):
        return self.text_detector(img)

    def detect_subtitle_batch(self, frames):
        return self.text_detector.batch(frames)

    def detect_subtitle_stream(self, batches):
        """iterable of frame batches -> generator of detect_subtitle_batch() results, next detectors in flight"""
        return self.text_detector.pipe.detect_stream(batches)


class TextDetector:
    def __init__(self, args):
        if getattr(args, "det_algorithm", "DB") != "DB":
            raise NotImplementedError("only DB detection exists in the reference (SURVEY F9)")
        model = _load_model(args.det_model_dir)
        ctx = _context()
        self.pipe = pipeline.OcrPipeline.__new__(pipeline.OcrPipeline)
        self.pipe.ctx = ctx
        self.pipe.det_weights = pipeline.resolve_det_weights(getattr(args, "det_weights", "auto"), model[1])
        self.pipe.det_input = "raw"
        self.pipe.det = engine.Net(ctx, model[0], model[1], fetch_cols=(0,), hilo=self.pipe.det_weights == "fp16x2",
                                   input_norm=pipeline.DET_NORM, fuse_preprocess=True)
        self.pipe.limit = getattr(args, "det_limit_side_len", 960)
        self.pipe.db = dict(thresh=getattr(args, "det_db_thresh", 0.3), box_thresh=getattr(args, "det_db_box_thresh", 0.6),
                            unclip_ratio=getattr(args, "det_db_unclip_ratio", 1.5))

    def __call__(self, img):
        t0 = time.time()
        boxes = self.pipe.detect(_to_device(img))[0]
        return np.asarray(boxes, dtype=np.float32).reshape(-1, 4, 2), time.time() - t0

    def batch(self, frames):
        return self.pipe.detect(frames)


class TextRecognizer:
    """paddleocr TextRecognizer(args)(img_list) -> (list[(text, score)], elapse): CTC recognition of ready-made crops
    (uint8 BGR arrays of any sizes), grouped like the reference (sorted by w/h, chunks of rec_batch_num)."""

    def __init__(self, args):
        model = _load_model(args.rec_model_dir)
        shp = [int(v) for v in getattr(args, "rec_image_shape", "3,48,320").split(",")]
        charset = charset_for(getattr(args, "lang", config.language), _ncls(model[0]),
                              getattr(args, "rec_char_dict_path", None), getattr(args, "use_space_char", True))
        ctx = _context()
        self.pipe = pipeline.OcrPipeline.__new__(pipeline.OcrPipeline)
        self.pipe.ctx = ctx
        self.pipe.rec = engine.Net(ctx, model[0], model[1], want_probs=False, ragged=True)
        self.pipe.charset = charset
        self.pipe.rec_batch_num = getattr(args, "rec_batch_num", 6)
        self.pipe.rec_h, self.pipe.rec_base_w = shp[1], shp[2]
        # every crop at the padded width of its reference chunk (sorted by w/h, chunks of rec_batch_num), chunks sharing launches
        self.pipe.rec_mode = getattr(args, "rec_mode", "ragged")
        self.pipe.bucket, self.pipe.batch_round, self.pipe.max_rec_batch, self.pipe.min_rec_group = 64, 1, 64, 0
        self.pipe.profile_sink = None
        self.pipe.rec_streams = 1

    def __call__(self, img_list):
        t0 = time.time()
        return self.pipe.recognize_crops(list(img_list)), time.time() - t0


class PaddleOCR:
    """PaddleOCR(**kwargs)(img, cls=False) -> (boxes, rec_res, time_dict) as used at backend/tools/ocr.py:27,91-113."""

    # kwargs of the reference's call (ocr.py:91-113) that select a backend / memory pool / algorithm label and have no
    # meaning here; anything else unknown is an error, not silently dropped
    _BACKEND_KWARGS = {"use_gpu", "gpu_mem", "gpu_id", "use_onnx", "onnx_providers", "max_batch_size", "det", "rec", "cls",
                       "ocr_version", "show_log", "use_mp", "total_process_num", "enable_mkldnn", "cpu_threads",
                       "use_tensorrt", "precision", "ir_optim", "cls_model_dir", "use_xpu", "use_npu", "use_mlu",
                       "det_limit_type", "benchmark", "warmup"}

    def __init__(self, det_model_dir=None, rec_model_dir=None, rec_batch_num=6, drop_score=0.5, lang="ch",
                 use_angle_cls=False, rec_image_shape="3,48,320", rec_mode="ragged", rec_char_dict_path=None,
                 use_space_char=True, det_algorithm="DB", rec_algorithm="CRNN", det_limit_side_len=960,
                 det_db_thresh=0.3, det_db_box_thresh=0.6, det_db_unclip_ratio=1.5, **backend):
        if use_angle_cls:
            raise NotImplementedError("angle classifier is never enabled by the reference (ocr.py:104)")
        if rec_mode not in ("ragged", "reference"):
            # "bucketed" (crops padded to bucket widths) does NOT give the reference's per-chunk padding and is kept on
            # pipeline.OcrPipeline for A/B measurements only; the reference-shaped call site offers the two identical modes
            raise ValueError(f"rec_mode={rec_mode!r}: 'ragged' (default) or 'reference' (same results, the reference's launch structure)")
        if det_algorithm != "DB":
            raise NotImplementedError("only DB detection exists in the reference (SURVEY F9)")
        if rec_algorithm not in ("CRNN", "SVTR_LCNet", "SVTR_HGNet"):
            raise NotImplementedError(f"rec_algorithm={rec_algorithm!r}: only the CTC recognisers of the reference exist")
        unknown = set(backend) - self._BACKEND_KWARGS
        if unknown:
            raise TypeError(f"PaddleOCR(): unsupported arguments {sorted(unknown)}")
        det = _load_model(det_model_dir)
        rec = _load_model(rec_model_dir)
        shp = [int(v) for v in rec_image_shape.split(",")]
        charset = charset_for(lang, _ncls(rec[0]), rec_char_dict_path, use_space_char)
        self.pipe = pipeline.OcrPipeline(_context(), det, rec, charset, rec_batch_num=rec_batch_num, rec_h=shp[1],
                                         rec_base_w=shp[2], drop_score=drop_score, rec_mode=rec_mode,
                                         limit_side_len=det_limit_side_len, db_thresh=det_db_thresh,
                                         db_box_thresh=det_db_box_thresh, db_unclip_ratio=det_db_unclip_ratio)

    def __call__(self, img, cls=False):
        t0 = time.time()
        boxes, res = self.pipe.ocr(_to_device(img))[0]
        return boxes, res, {"all": time.time() - t0}

    def batch(self, frames):
        return self.pipe.ocr(frames)

    def stream(self, batches):
        """iterable of frame batches -> generator of batch() results, the detectors of the next batches already in flight
        (OcrPipeline.ocr_stream) and — in the ragged mode, where grouping never changes a result — the crops of two consecutive
        batches sharing the recogniser's launch sequences."""
        return self.pipe.ocr_stream(batches, rec_span=2)


class OcrRecogniser:
    """backend/tools/ocr.py:9-113."""

    def __init__(self):
        self.recogniser = None
        self.hardware_accelerator = HardwareAccelerator()

    @staticmethod
    def y_round(y):
        y_min = y + 10 - y % 10
        y_max = y - y % 10
        return y_min if abs(y - y_min) < abs(y - y_max) else y_max

    def init_model(self):
        mc = PaddleModelConfig(self.hardware_accelerator)
        hw = self.hardware_accelerator
        return PaddleOCR(use_gpu=hw.has_cuda(), gpu_mem=500, det_algorithm='DB', det_model_dir=mc.DET_MODEL_PATH,
                         rec_algorithm='CRNN', rec_batch_num=config.recBatchNumber, rec_model_dir=mc.REC_MODEL_PATH,
                         max_batch_size=config.maxBatchSize, det=True, use_angle_cls=False, drop_score=0,
                         lang=mc.REC_CHAR_TYPE, ocr_version=f'PP-OCR{mc.MODEL_VERSION.lower()}',
                         rec_image_shape=mc.REC_IMAGE_SHAPE, use_onnx=False, onnx_providers=hw.onnx_providers)

    def predict(self, image):
        if not self.recogniser:
            self.recogniser = self.init_model()
        detection_box, recognise_result, _ = self.recogniser(image, cls=False)
        return self.arrange(detection_box, recognise_result)

    def predict_batch(self, frames):
        if not self.recogniser:
            self.recogniser = self.init_model()
        return [self.arrange(b, r) for b, r in self.recogniser.batch(frames)]

    def predict_with_dets(self, frames, dets):
        """predict_batch(frames) for frames whose detector output `dets` is already known (accurate mode: SubtitleDetect ran
        on them a moment ago with the same detector)."""
        if not self.recogniser:
            self.recogniser = self.init_model()
        return [self.arrange(b, r) for b, r in self.recogniser.pipe.ocr_from_det(frames, dets)]

    def predict_stream(self, batches):
        """iterable of frame batches (device uint8 [n,H,W,3]) -> generator of predict_batch() results in order, with the
        detector of the following batches overlapping the recognition of the current one."""
        if not self.recogniser:
            self.recogniser = self.init_model()
        for out in self.recogniser.stream(batches):
            yield [self.arrange(b, r) for b, r in out]

    @classmethod
    def arrange(cls, detection_box, recognise_result):
        """Line grouping / ordering of ocr.py:28-86 (a3)."""
        if len(detection_box) == 0:
            return detection_box, recognise_result
        coords = [list(_aabb(q)) for q in detection_box] if isinstance(detection_box, list) else []
        lines = []
        for c in coords:
            yr = cls.y_round(c[2])
            if len(lines) < 1 or (yr not in lines and yr + 10 not in lines and yr - 10 not in lines):
                lines.append(yr)
        lines = sorted(lines)
        for c in coords:
            for j in lines:
                if abs(j - cls.y_round(c[2])) <= 10:
                    c[2] = j
        pairs = list(zip(coords, recognise_result))
        ranked = []
        for line in lines:
            row = [p for p in pairs if p[0][2] == line]
            for l in range(1, len(row)):             # bubble sort by xmin, like the reference (stable, O(n^2))
                for j in range(0, len(row) - l):
                    if row[j][0][0] > row[j + 1][0][0]:
                        row[j], row[j + 1] = row[j + 1], row[j]
            ranked += row
        dt_box = [[(c[0], c[2]), (c[1], c[2]), (c[1], c[3]), (c[0], c[3])] for c, _ in ranked]
        return dt_box, [r for _, r in ranked]


def _aabb(q):
    q = list(q)
    x1, y1 = int(q[0][0]), int(q[0][1])
    x2, y2 = int(q[1][0]), int(q[1][1])
    x3, y3 = int(q[2][0]), int(q[2][1])
    x4, y4 = int(q[3][0]), int(q[3][1])
    return max(x1, x4), min(x2, x3), max(y1, y2), min(y3, y4)


def get_coordinates(dt_box):
    """backend/tools/ocr.py:115-134 — note: returns [] unless `dt_box` is a python list (callers pass .tolist())."""
    out = []
    if isinstance(dt_box, list):
        for q in dt_box:
            out.append(_aabb(q))
    return out


def subtitle_area_keep(coordinate, prob, sub_area, deviation_rate=None, drop_score=None):
    """Area / confidence filter of extract_subtitles (backend/tools/subtitle_ocr.py:42-67) on axis-aligned rectangles.
    coordinate = (xmin,xmax,ymin,ymax); sub_area has .xmin .xmax .ymin .ymax."""
    deviation_rate = config.subtitleAreaDeviationRate if deviation_rate is None else deviation_rate
    drop_score = config.dropScore / 100.0 if drop_score is None else drop_score
    # a skewed quad can give xmin > xmax or ymin > ymax (max of the left corners vs min of the right ones, ocr.py:118-129); the
    # reference builds a shapely Polygon from the four corners, whose region does not depend on the corner order — so the
    # rectangle is taken between the smaller and the larger value (found by fuzzing against the reference's own code)
    xmin, xmax = min(coordinate[0], coordinate[1]), max(coordinate[0], coordinate[1])
    ymin, ymax = min(coordinate[2], coordinate[3]), max(coordinate[2], coordinate[3])
    ax0, ax1 = min(sub_area.xmin, sub_area.xmax), max(sub_area.xmin, sub_area.xmax)
    ay0, ay1 = min(sub_area.ymin, sub_area.ymax), max(sub_area.ymin, sub_area.ymax)
    ix0, ix1 = max(xmin, ax0), min(xmax, ax1)
    iy0, iy1 = max(ymin, ay0), min(ymax, ay1)
    if ix0 > ix1 or iy0 > iy1:
        return False
    inter = max(0, ix1 - ix0) * max(0, iy1 - iy0)
    a_area = (ax1 - ax0) * (ay1 - ay0)
    b_area = (xmax - xmin) * (ymax - ymin)
    return (a_area + b_area - inter) / a_area - 1 <= deviation_rate and prob > drop_score


def extract_subtitles(frame_no, predict_result, sub_area=None, rec_char_type=None, deviation_rate=None, drop_score=None):
    """Raw subtitle lines of one frame as extract_subtitles writes them (backend/tools/subtitle_ocr.py:20-85):
    '%08d\t(xmin, xmax, ymin, ymax)\ttext\n' for every recognised line that passes the area / confidence filter
    (all lines when no area is given); CJK ideographs are stripped first when the language is 'en' (:35-37)."""
    import re
    dt_box, rec_res = predict_result
    lang = config.language if rec_char_type is None else rec_char_type
    lines = []
    for (text, prob), coordinate in zip(rec_res, get_coordinates(dt_box)):
        if lang == 'en':
            text = re.sub('[\u4e00-\u9fa5]', '', text)
        if sub_area is None or subtitle_area_keep(coordinate, prob, sub_area, deviation_rate, drop_score):
            lines.append(f'{str(frame_no).zfill(8)}\t{coordinate}\t{text}\n')
    return lines
