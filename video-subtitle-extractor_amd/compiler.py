"""Graph compiler: model descriptor (+ fp32 weights) -> fused NHWC/fp16 engine program.

Host-side only (numpy).  The output `Program` is what the C-ABI runtime executes: a flat array of
`vse_op` records (ir.OP_DT), a packed weight blob and a workspace size.  Design points (MI355X-first,
not a translation of the Paddle executor):

  * activations are NHWC fp16 with channels padded to a multiple of 8, so every implicit-GEMM gather is
    a 16-byte vector per (tap, 8-channel group); weights are pre-tiled [K/64][Cout][64] fp16 in exactly
    the order the conv kernel streams them through LDS;
  * batch-norm, conv bias and the PP-LCNetV3 "learnable affine" scalars before the activation are folded
    into the weights/bias in fp32 at compile time; activation, post-activation affine, residual add
    (optionally nearest-upsampled: FPN top-down adds) run in the conv epilogue;
  * channel concatenation is free: producers write straight into channel slices of the concat buffer
    (a view = buffer + channel offset + pixel stride);
  * layout-only ops (flatten/transpose/reshape/squeeze between NCHW and [B,T,C]) vanish because
    NHWC [B,1,T,C] *is* [B,T,C];
  * SVTR attention sub-graphs are pattern-matched into one fused attention op; final class softmax is one
    op that also emits argmax + max-prob for the CTC collapse;
  * buffers are placed in one workspace arena by liveness (first-fit), sized for the batch.

Reference call sites this replaces: the Paddle predictor `run()` inside paddleocr's TextDetector /
TextRecognizer, reached from backend/tools/subtitle_detect.py:25 and backend/tools/ocr.py:27.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import os
import numpy as np

from . import ir
from .ir import dev_switch as _dev_switch
from .chains import ChainMixin


def rup(x, m):
    return (x + m - 1) // m * m


# ------------------------------------------------------------------------------------------------ views
@dataclass
class Buf:
    id: int
    n: int
    h: int
    w: int
    ld: int
    esize: int = 2
    ext: Optional[int] = None      # external slot index (0 = input, 1.. = outputs) or None
    first: int = 1 << 30
    last: int = -1
    offset: int = 0
    wl: Optional[int] = None       # ragged plans: width level of the tensor (index into Program.wlevels); None = no width
    lo_off: int = 0                # hi + lo PAIR tensor: channels [lo_off, 2 lo_off) of every pixel hold fp16(x - fp16(x)); 0 = plain fp16

    @property
    def nbytes(self):
        return self.n * self.h * self.w * self.ld * self.esize


@dataclass
class View:
    buf: Buf
    coff: int                      # channel offset inside the buffer
    n: int
    h: int
    w: int
    segs: List[Tuple[int, int]]    # [(physical start relative to coff, logical count)]
    span: int                      # physical channel span
    up: int = 0                    # virtual nearest-upsample shift (logical h,w already upsampled)
    tag: str = "nchw"              # logical layout of the Paddle tensor this view stands for
    parts: Optional[list] = None   # virtual 2-way concat: [View, View] gathered by the consumer conv (no copy)
    dense1: bool = False           # ONE channel stored densely (buffer ld = 1) behind a nominal 8-channel span: only F_UP2HEAD reads it (F_TAIL2)
    gate: Optional[tuple] = None   # (gate View, flags): the tensor is  this * gate  (F_RES: this * (1 + gate)), not yet multiplied — an SE
                                   # output whose only reader is a concat: the copy into the slot applies it (lower_concat)

    @property
    def c(self):
        return sum(s[1] for s in self.segs)

    def chmap(self):
        out = []
        for st, cnt in self.segs:
            out += list(range(st, st + cnt))
        return np.asarray(out, dtype=np.int64)

    @property
    def src_h(self):
        return self.h >> self.up

    @property
    def src_w(self):
        return self.w >> self.up


def merge_segs(segs):
    """Adjacent channel segments that are physically contiguous are one segment (a concat of 8-multiples is dense)."""
    out = []
    for st, cnt in segs:
        if out and out[-1][0] + out[-1][1] == st:
            out[-1] = (out[-1][0], out[-1][1] + cnt)
        else:
            out.append((st, cnt))
    return out


@dataclass
class Program:
    ops: np.ndarray                 # ir.OP_DT records
    weights: "WeightStore"          # shared packed weight blob
    ws_bytes: int
    in_shape: Tuple[int, int, int, int]          # N,H,W,Cphys of the fp16 NHWC input
    outputs: List[dict] = field(default_factory=list)   # [{name, n,h,w,c,ld,esize,kind}]
    names: List[str] = field(default_factory=list)      # debug: op -> originating tensor name
    gmacs: float = 0.0              # algorithmic MACs of conv/linear ops (for the roofline)
    op_gmacs: List[float] = field(default_factory=list)   # per op, aligned with `ops`
    # ragged plans (compile_model(ragged=True)): level 0 = the input width of a sample; level l = (parent, k, s, p, ceil):
    # width = the output width of a window of k, stride s, padding p over the parent's width.  width_table() evaluates it.
    wlevels: Optional[list] = None
    out_level: int = 0              # level of the sequence the class softmax runs over (T of a sample)

    def width_table(self, widths):
        """int32 [levels][N]: per-sample width of every level for input widths `widths` (len N)."""
        assert self.wlevels is not None, "not a ragged plan"
        w0 = np.asarray(widths, dtype=np.int64).reshape(-1)
        assert w0.shape[0] == self.in_shape[0] and w0.min() >= 1 and w0.max() <= self.in_shape[2], (w0, self.in_shape)
        tab = np.zeros((len(self.wlevels), w0.shape[0]), np.int64)
        tab[0] = w0
        for l in range(1, len(self.wlevels)):
            parent, k, s, p, ceil = self.wlevels[l]
            tab[l] = level_width(tab[parent], k, s, p, ceil)
        return tab.astype(np.int32)


def level_width(w, k, s, p, ceil):
    """Output width of a k-wide window with stride s and padding p over width w (numpy array or int); Paddle's pool2d
    ceil_mode rule when `ceil` (the last window must start inside the input or its left padding)."""
    if not ceil:
        return (w + 2 * p - k) // s + 1
    o = -(-(w + 2 * p - k) // s) + 1
    return np.where((o - 1) * s >= w + p, o - 1, o) if isinstance(o, np.ndarray) else (o - 1 if (o - 1) * s >= w + p else o)


class WeightStore:
    """Packed weight blob shared by every plan compiled for one model (offsets are shape-independent)."""

    def __init__(self):
        self.blob = bytearray()
        self.index = {}

    def add(self, key, arr):
        if key in self.index:
            return self.index[key]
        if callable(arr):
            arr = arr()
        off = rup(len(self.blob), 256)
        self.blob.extend(b"\0" * (off - len(self.blob)))
        self.blob.extend(np.ascontiguousarray(arr).tobytes())
        self.index[key] = off
        return off

    def array(self):
        return np.frombuffer(bytes(self.blob), dtype=np.uint8).copy()

    def snapshot(self):
        """State to roll back to when a compile attempt is abandoned (compile_model's fallbacks): the blobs an aborted attempt
        packed would otherwise stay in the store — uploaded with every plan, read by none."""
        return len(self.blob), dict(self.index)

    def rollback(self, snap):
        n, index = snap
        del self.blob[n:]
        self.index = dict(index)          # a copy: the snapshot must survive the adds that follow (it may be rolled back to again)


# ------------------------------------------------------------------------------------------------ compiler
_VIRTUAL = {"nearest_interp_v2", "flatten_contiguous_range", "transpose2", "reshape2", "squeeze2", "dropout",
            "assign", "shape", "fill_constant", "fill_constant_batch_size_like", "scale_noop"}
_ACTS = {"relu": ir.ACT_RELU, "hard_swish": ir.ACT_HSWISH, "swish": ir.ACT_SWISH, "sigmoid": ir.ACT_SIGMOID,
         "hard_sigmoid": ir.ACT_HSIGMOID}


# shortest K (taps x channels) worth a conv_patch_kernel launch; VSE_PATCH_MINK overrides it for kernel experiments
PATCH_MIN_K = int(_dev_switch("VSE_PATCH_MINK", "500"))
# most couts sent to the patch kernel: with more than 64 couts the 256-pixel implicit-GEMM tiles (conv_gemm.hip,
# activation tile fetched once for 128-256 couts) measure 15-50 % faster than the patch kernel on MI355X
PATCH_MAX_COUT = 64
# smallest useful fraction of a tile grid (8/16 x 32 output pixels) for the patch kernel; below it the map is too ragged
PATCH_MIN_TILE_EFF = float(_dev_switch("VSE_PATCH_MINEFF", "0.5"))
# LIGHT patch variant policy, mirrors VSE_PATCH_LIGHT in csrc/conv_patch.hip: 0 never, 1 layers with 65-128 couts, 2 all eligible
PATCH_LIGHT = int(_dev_switch("VSE_PATCH_LIGHT", "2"))
# evaluate the PP-OCRv4 server detector's last 3x3 conv on the low-res grid (conv_head.hip); VSE_HEAD_UP2=0 keeps it on
# the patch kernel (experiments / A-B)
STEM = _dev_switch("VSE_STEM", "1") != "0"         # conv_stem_kernel for 3x3 convs over <= 4 real channels
GATE_DW = _dev_switch("VSE_GATE_DW", "1") != "0"   # SE gate folded into a depthwise consumer
GATE_FOLD = _dev_switch("VSE_GATE_FOLD", "1") != "0"   # SE gate whose consumers are a depthwise conv and 1x1 convs: folded into them
WK32 = _dev_switch("VSE_WK32", "1") != "0"      # 32-deep weight tiles for conv_gemm_kernel (contiguous wave DMAs)
HEAD_UP2 = _dev_switch("VSE_HEAD_UP2", "1") != "0"
COL = _dev_switch("VSE_COL", "1") != "0"           # conv_col_kernel (one filter column per step) for 9x9 / 7x7 / 5x5 layers
# below this tile efficiency the 8-row tiles of conv_patch_kernel win (measured: 17x30 map 0.163 vs 0.203 ms, 34x60 0.50 vs 0.43)
COL_MIN_TILE_EFF = float(_dev_switch("VSE_COL_MINEFF", "0.75"))
# conv_c3_kernel (3x3, two blocks per CU): VSE_COL3=0 off; cout / tile-efficiency limits from per-layer A/B runs
COL3 = _dev_switch("VSE_COL3", "1") != "0"
COL3_MAX_COUT = int(_dev_switch("VSE_COL3_MAXCOUT", "192"))     # per-layer A/B (tools/bench_conv.py --cfgs d,p,c): 224-cout layers tie or lose
PW = _dev_switch("VSE_PW", "1") != "0"             # conv_pw_kernel for 1x1 convs (and 2x2 s2 transposed convs) over <= 64 input channels
PW_MAX_COUT = int(_dev_switch("VSE_PW_MAXCOUT", "64"))
TAIL2 = _dev_switch("VSE_TAIL2", "1") != "0"       # the server detector's second head deconv inside the first one's launch (F_TAIL2)
COL3_MIN_K = int(_dev_switch("VSE_COL3_MINK", "250"))   # 3x3 32->32 @136x240 (K = 288): 0.180 ms on the implicit GEMM, 0.115 ms here
COL3_WIDE_MIN_CIN = 128    # layers with more than 64 couts (two+ cout tiles refetch the patch) only from 128 input channels on
COL3_MIN_TILE_EFF = float(_dev_switch("VSE_COL3_MINEFF", "0.8"))
# nominal sample width for the kernel selection of ragged (recogniser) plans: subtitle lines are several hundred pixels wide at
# 48 px height; every plan of a model selects as if its maps were this wide (the choice only steers efficiency, never results
# ACROSS plans of one process; a different value is a different set of summation orders)
RAGGED_SEL_W = int(_dev_switch("VSE_RAGGED_SELW", "768"))
ONECH = _dev_switch("VSE_ONECH", "1") != "0"             # DB head: last transposed conv stores the fp32 map directly
GATE_CONCAT = _dev_switch("VSE_GATE_CONCAT", "1") != "0"   # SE output that only feeds a concat: multiplied by the copy into the slot
PAIR_MAX_PIX = int(_dev_switch("VSE_PAIR_MAXPIX", str(1 << 40)))     # experiments: tensors with more pixels per image stay plain fp16
PAIR_MIN_PIX = int(_dev_switch("VSE_PAIR_MINPIX", "0"))                # ... and tensors with fewer
DWPW = _dev_switch("VSE_DWPW", "1") != "0"                 # depthwise conv fused in front of its 1x1 consumer (hi + lo nets: conv_dwpw.hip)
# filter sizes sent there: 3x3 wins against depthwise + 1x1 launches (V4 16 -> 32 @272x480: 0.33 vs 0.44 ms), 5x5 loses (V3 64 -> 24
# @68x120: 0.21 vs 0.14 ms: 25 taps of fp32 VALU work per 8 channels and lane, no window sharing between neighbouring pixels)
# (5 x 5 filters lose fused — 25 taps per lane — and their kernel instantiations exist in development builds only)
DWPW_K = tuple(int(v) for v in _dev_switch("VSE_DWPW_K", "3").split(",")) if os.environ.get("VSE_DEV_BUILD", "0") == "1" else (3,)
HLSUM = _dev_switch("VSE_HLSUM", "1") != "0"              # 3x3 convs with <= 32 couts of a hi + lo net: hi | lo weight rows in one pass (F_HLSUM)
SE_LATERAL = _dev_switch("VSE_SE_LATERAL", "1") != "0"   # 1x1 conv + SE block with shortcut -> one gated conv (F_OGATE)
LSTM_MFMA = _dev_switch("VSE_LSTM_MFMA", "1") != "0"     # batch-shared MFMA recurrence (csrc/lstm.hip) for 256-unit LSTMs
LSTM_WAVES = int(_dev_switch("VSE_LSTM_WAVES", "16"))    # 8: lstm_mfma_kernel, 16: lstm_mfma16_kernel (twice the loads in flight)


def c3_tile_eff(oh, ow):
    """Mirror of conv_c3_plan (csrc/conv_c3.hip): best tile efficiency over the 16x32 / 8x64 / 4x128 tile shapes when waves
    outside the map idle."""
    def axis(n, unit, waves):
        tile = unit * waves
        full, rem = divmod(n, tile)
        return full + ((0.35 + 0.65 * -(-rem // unit) / waves) if rem else 0.0)
    return max(oh * ow / (axis(oh, 2, rw) * axis(ow, 32, 8 // rw) * 512.0) for rw in (8, 4, 2))


class UnsupportedGraph(NotImplementedError):
    """A descriptor uses an attribute value the kernels hard-code differently (would compile and compute wrong values)."""


class GatedConvUnsupported(UnsupportedGraph):
    """A conv rewritten by Compiler._rewrite_se_laterals (F_OGATE) met a layer form its epilogue cannot express: compile_model
    retries the graph WITHOUT that rewrite (the rewrite itself cannot be undone on a half-lowered graph)."""


class Tail2Unsupported(UnsupportedGraph):
    """The dense 1-channel map an F_TAIL2 conv writes (ld = 1) reached a consumer other than the F_UP2HEAD conv that reads it at pixel
    stride: compile_model retries the graph with the two transposed convs as separate launches."""


# Attribute values the lowering hard-codes (every graph under backend/models/ satisfies them).  A descriptor
# converted from another export (SAME padding, dilated convs, align_corners resize ...) must fail here, loudly, instead of
# compiling into something that silently computes different values.  A missing attribute means the Paddle default.
_REQUIRED_ATTRS = {
    "conv2d": {"dilations": [1, 1], "padding_algorithm": "EXPLICIT", "data_format": ("NCHW", "AnyLayout")},
    "depthwise_conv2d": {"dilations": [1, 1], "padding_algorithm": "EXPLICIT", "data_format": ("NCHW", "AnyLayout")},
    "conv2d_transpose": {"dilations": [1, 1], "padding_algorithm": "EXPLICIT", "data_format": ("NCHW", "AnyLayout"),
                         "output_padding": [], "output_size": []},
    "hard_swish": {"offset": 3.0, "scale": 6.0, "threshold": 6.0},
    "swish": {"beta": 1.0},
    "nearest_interp_v2": {"align_corners": False, "interp_method": "nearest", "data_layout": ("NCHW", "AnyLayout")},
    "pool2d": {"padding_algorithm": "EXPLICIT", "global_pooling": False, "data_format": ("NCHW", "AnyLayout")},
    "batch_norm": {"data_layout": ("NCHW", "AnyLayout")},
    "matmul_v2": {"trans_x": False, "trans_y": False},
}


def check_attrs(ops):
    for i, op in enumerate(ops):
        want = _REQUIRED_ATTRS.get(op["type"])
        if not want:
            continue
        a = op.get("attrs", {})
        for k, v in want.items():
            if k not in a:
                continue
            got = a[k]
            if v == [] and isinstance(got, (list, tuple)) and not any(got):
                continue        # output_padding / output_size written out as explicit zeros = the default
            ok = got in v if isinstance(v, tuple) else (abs(got - v) < 1e-6 if isinstance(v, float) else got == v)
            if not ok:
                raise UnsupportedGraph(f"op {i} ({op['type']}): attribute {k}={got!r} is not supported (the kernels "
                                       f"implement {k}={v!r} only)")


class Compiler(ChainMixin):
    def __init__(self, desc, weights, batch, height, width, fetch_cols=(0,), want_probs=True,
                 store=None, reuse=True, se_lateral=None, tail2=None):
        self.desc = desc
        self.tail2 = tail2                    # False: never fuse the head's second transposed conv into the first (F_TAIL2)
        self.W = dict(weights)
        self.ops = list(desc["ops"])
        check_attrs(self.ops)
        self.merged_gmac_credit = {}          # merged conv weight name -> algorithmic MAC factor of the original branches
        self.hilo = False                     # fp16 hi + lo weight pairs (compile_model(hilo=True))
        self.pending_gate = {}                # SE output name -> gate view its depthwise consumer applies on load (F_GATE)
        self.pending_wgate = {}               # SE output name -> gate view its 1x1 conv consumers fold into per-image weights (F_IMGW)
        # ragged plans (recognisers): every sample of the batch carries its own width; see ir.P_WLIN / Program.wlevels
        self.ragged = False
        self.wlevels = [None]                 # level 0 = the input width
        self.wlevel_index = {}
        self.sel_w0 = RAGGED_SEL_W            # kernel selection of a ragged plan looks at THIS input width, never at the batch's
        self.input_norm = None                # (mean3, std3): see fold_input_norm
        self.fuse_preprocess = False          # with input_norm: the stem conv reads the uint8 frames and resizes them itself (F_U8SRC)
        self._merge_parallel_convs()
        if SE_LATERAL if se_lateral is None else se_lateral:
            self._rewrite_se_laterals()
        self.N, self.H, self.Wd = batch, height, width
        self.fetch_cols = tuple(fetch_cols)
        self.want_probs = want_probs
        self.env: Dict[str, View] = {}
        self.bufs: List[Buf] = []
        self.ir_ops: List[dict] = []
        self.store = store if store is not None else WeightStore()
        self.reuse = reuse
        self.use_patch = True
        self.use_col = True                   # conv_col_kernel / conv_c3_kernel (they walk hi + lo weights; the patch kernel does not)
        self.done = set()
        self.outputs = []
        self.gmacs = 0.0
        # producer / consumer maps
        self.producer = {}
        self.consumers = {}
        for i, op in enumerate(self.ops):
            for outs in op["out"].values():
                for o in outs:
                    self.producer.setdefault(o, i)
            for ins in op["in"].values():
                for n in ins:
                    self.consumers.setdefault(n, []).append(i)
        self.fetched_names = {n for op in self.ops if op["type"] == "fetch" for n in op["in"].get("X", [])}
        self._mark_live()
        self._plan_concats()

    # -------------------------------------------------------------------------------------------- graph rewrite
    def _merge_parallel_convs(self):
        """Re-parameterisation: conv_a(T) + conv_b(T) (both stride 1, 'same' padding, own bias, no activation in
        between) == ONE conv with the two kernels embedded, centred, in a common (max kh) x (max kw) window and the
        biases added.  PP-OCRv4's IntraCL blocks sum a k x k, a k x 1 and a 1 x k conv of the same tensor: three
        launches + two adds become one k x k conv (the k x 1 / 1 x k taps ride along for free on the matrix cores).
        Exact in real arithmetic; applied repeatedly until nothing matches."""
        changed = True
        while changed:
            changed = False
            prod, cons = {}, {}
            for i, op in enumerate(self.ops):
                for outs in op["out"].values():
                    for o in outs:
                        prod.setdefault(o, i)
                for ins in op["in"].values():
                    for n in ins:
                        cons.setdefault(n, []).append(i)

            def branch(name):
                """name = output of [conv2d -> elementwise_add(bias)] with single consumers -> (i_conv, i_bias) or None."""
                ib = prod.get(name)
                if ib is None or self.ops[ib]["type"] != "elementwise_add" or len(cons.get(name, [])) != 1:
                    return None
                b = self.ops[ib]
                y = b["in"]["Y"][0]
                if y not in self.W or b["attrs"].get("axis", -1) != 1:
                    return None
                ic = prod.get(b["in"]["X"][0])
                if ic is None or self.ops[ic]["type"] != "conv2d" or len(cons.get(b["in"]["X"][0], [])) != 1:
                    return None
                c = self.ops[ic]
                a = c["attrs"]
                w = self.W[c["in"]["Filter"][0]]
                pads = a["paddings"]
                if list(a["strides"]) != [1, 1] or a.get("groups", 1) != 1 or len(pads) != 2:
                    return None
                if w.shape[2] % 2 == 0 or w.shape[3] % 2 == 0 or pads[0] != w.shape[2] // 2 or pads[1] != w.shape[3] // 2:
                    return None
                return ic, ib

            for i, op in enumerate(self.ops):
                if op["type"] != "elementwise_add" or op["attrs"].get("axis", -1) not in (-1,):
                    continue
                xa, xb = op["in"]["X"][0], op["in"]["Y"][0]
                if xa in self.W or xb in self.W:
                    continue
                ba, bb = branch(xa), branch(xb)
                if ba is None or bb is None:
                    continue
                ca, cb = self.ops[ba[0]], self.ops[bb[0]]
                if ca["in"]["Input"][0] != cb["in"]["Input"][0]:
                    continue
                wa, wb = self.W[ca["in"]["Filter"][0]], self.W[cb["in"]["Filter"][0]]
                if wa.shape[:2] != wb.shape[:2]:
                    continue
                kh, kw = max(wa.shape[2], wb.shape[2]), max(wa.shape[3], wb.shape[3])
                wm = np.zeros(wa.shape[:2] + (kh, kw), np.float64)
                for w_ in (wa, wb):
                    oy, ox = (kh - w_.shape[2]) // 2, (kw - w_.shape[3]) // 2
                    wm[:, :, oy:oy + w_.shape[2], ox:ox + w_.shape[3]] += w_
                wname = ca["in"]["Filter"][0] + "+" + cb["in"]["Filter"][0]
                bname = self.ops[ba[1]]["in"]["Y"][0] + "+" + self.ops[bb[1]]["in"]["Y"][0]
                self.W[wname] = wm.astype(np.float32)
                self.W[bname] = (self.W[self.ops[ba[1]]["in"]["Y"][0]].astype(np.float64) +
                                 self.W[self.ops[bb[1]]["in"]["Y"][0]].astype(np.float64)).astype(np.float32)
                # algorithmic MACs stay those of the ORIGINAL branches (the roofline counts the reference's work)
                taps = lambda n, w_: self.merged_gmac_credit.get(n, w_.shape[2] * w_.shape[3])
                self.merged_gmac_credit[wname] = taps(ca["in"]["Filter"][0], wa) + taps(cb["in"]["Filter"][0], wb)
                out = op["out"]["Out"][0]
                mid = out + ":merged_conv"
                conv = {"type": "conv2d", "in": {"Input": [ca["in"]["Input"][0]], "Filter": [wname]},
                        "out": {"Output": [mid]},
                        "attrs": {"strides": [1, 1], "paddings": [kh // 2, kw // 2], "dilations": [1, 1], "groups": 1}}
                bias = {"type": "elementwise_add", "in": {"X": [mid], "Y": [bname]}, "out": {"Out": [out]},
                        "attrs": {"axis": 1}}
                drop = {ba[0], ba[1], bb[0], bb[1], i}
                first = min(drop)
                new_ops = []
                for k, o in enumerate(self.ops):
                    if k == first:
                        new_ops += [conv, bias]
                    if k not in drop:
                        new_ops.append(o)
                # the merged conv must come after its input is produced: `first` is the earlier branch conv, whose
                # input precedes it already
                self.ops = new_ops
                changed = True
                break

    def _rewrite_se_laterals(self):
        """x = conv1x1(c);  o = x + x * hsigmoid(fc2(relu(fc1(avgpool(x)))))   (PaddleOCR RSELayer with shortcut, the laterals of
        the mobile detectors' RSE-FPN: 96 channels at up to 136 x 240 from a 12..56-channel input)
        ->  avgpool(x) = conv1x1(avgpool(c)) — a 1x1 conv without padding commutes with the spatial mean —, so the gate is computed
        from the mean of the NARROW input, and  o = conv1x1(c) * (1 + gate)  is ONE conv with an output gate in its epilogue
        (F_OGATE).  x (the widest tensor of the neck), its pooling pass, the multiply and the add are never executed; the top-down
        add behind o then rides in the same epilogue as a residual.  Exact in real arithmetic."""
        changed = True
        while changed:
            changed = False
            prod, cons = {}, {}
            for i, op in enumerate(self.ops):
                for outs in op["out"].values():
                    for o in outs:
                        prod.setdefault(o, i)
                for ins in op["in"].values():
                    for nme in ins:
                        cons.setdefault(nme, []).append(i)

            def single(name, typ):
                c = cons.get(name, [])
                return c[0] if len(c) == 1 and self.ops[c[0]]["type"] == typ else None
            for ic, op in enumerate(self.ops):
                if op["type"] != "conv2d" or "out_gate" in op["attrs"]:
                    continue
                a = op["attrs"]
                w = self.W.get(op["in"]["Filter"][0])
                if (w is None or tuple(w.shape[2:]) != (1, 1) or list(a["strides"]) != [1, 1] or any(a["paddings"]) or a.get("groups", 1) != 1
                        or w.shape[0] % 8):
                    continue
                x = op["out"]["Output"][0]
                cx = cons.get(x, [])
                if len(cx) != 3 or sorted(self.ops[j]["type"] for j in cx) != ["elementwise_add", "elementwise_mul", "pool2d"]:
                    continue
                ipool = next(j for j in cx if self.ops[j]["type"] == "pool2d")
                imul = next(j for j in cx if self.ops[j]["type"] == "elementwise_mul")
                iadd = next(j for j in cx if self.ops[j]["type"] == "elementwise_add")
                pa = self.ops[ipool]["attrs"]
                if not (pa.get("pooling_type") == "avg" and (pa.get("adaptive", False) and list(pa.get("ksize", [])) == [1, 1] or pa.get("global_pooling", False))):
                    continue
                # pooled -> conv (+bias) -> relu -> conv (+bias) -> hard_sigmoid -> gate
                nme = self.ops[ipool]["out"]["Out"][0]
                ok = True
                for typ in ("conv2d", "elementwise_add", "relu", "conv2d", "elementwise_add", "hard_sigmoid"):
                    j = single(nme, typ)
                    if j is None:
                        ok = False
                        break
                    o2 = self.ops[j]
                    nme = (o2["out"].get("Output") or o2["out"].get("Out"))[0]
                if not ok:
                    continue
                gate = nme
                mul, add = self.ops[imul], self.ops[iadd]
                if sorted([mul["in"]["X"][0], mul["in"]["Y"][0]]) != sorted([x, gate]) or cons.get(gate, []) != [imul]:
                    continue
                m = mul["out"]["Out"][0]
                if sorted([add["in"]["X"][0], add["in"]["Y"][0]]) != sorted([x, m]) or cons.get(m, []) != [iadd]:
                    continue
                cin_name = op["in"]["Input"][0]
                pc = x + ":gap_of_input"
                pool_c = {"type": "pool2d", "in": {"X": [cin_name]}, "out": {"Out": [pc]}, "attrs": dict(pa)}
                conv_p = {"type": "conv2d", "in": {"Input": [pc], "Filter": list(op["in"]["Filter"])},
                          "out": {"Output": [self.ops[ipool]["out"]["Out"][0]]}, "attrs": dict(a)}
                gated = {"type": "conv2d", "in": dict(op["in"], Gate=[gate]), "out": {"Output": [add["out"]["Out"][0]]},
                         "attrs": dict(a, out_gate=gate)}          # (the gate is an INPUT: liveness and ordering see it)
                new_ops = []
                for k, o in enumerate(self.ops):
                    if k == ic:
                        new_ops += [pool_c, conv_p]
                    elif k == iadd:
                        new_ops.append(gated)
                    elif k in (ipool, imul):
                        continue
                    else:
                        new_ops.append(o)
                self.ops = new_ops
                changed = True
                break

    # -------------------------------------------------------------------------------------------- helpers
    def _mark_live(self):
        """Dead-code elimination backwards from the requested fetch columns."""
        need = set()
        for op in self.ops:
            if op["type"] == "fetch" and op["attrs"].get("col", 0) in self.fetch_cols:
                need.add(op["in"]["X"][0])
        self.live = [False] * len(self.ops)
        for i in range(len(self.ops) - 1, -1, -1):
            op = self.ops[i]
            if op["type"] == "fetch":
                self.live[i] = op["attrs"].get("col", 0) in self.fetch_cols
                continue
            outs = [o for v in op["out"].values() for o in v]
            if any(o in need for o in outs):
                self.live[i] = True
                for v in op["in"].values():
                    need.update(v)

    def _live_consumers(self, name):
        return [i for i in self.consumers.get(name, []) if self.live[i]]

    def _plan_concats(self):
        """tensor name -> (concat out name, physical channel offset) so producers write in place."""
        self.placement = {}
        self.concat_layout = {}
        self.virtual_concats = set()
        for i, op in enumerate(self.ops):
            if op["type"] != "concat" or not self.live[i]:
                continue
            out = op["out"]["Out"][0]
            self.concat_layout[out] = None   # resolved lazily when channel counts are known
            if self._concat_is_virtual(i):
                self.virtual_concats.add(out)
                continue
            for name in op["in"]["X"]:
                if name not in self.placement and len([c for c in self.consumers.get(name, [])
                                                        if self.ops[c]["type"] == "concat"]) == 1:
                    self.placement[name] = out

    def _concat_is_virtual(self, i):
        """2-input concat with a nearest-upsampled input whose ONLY consumer is a k x k stride-1 conv: the patch conv
        kernel gathers both sources itself (second source with its own shift), nothing is copied."""
        op = self.ops[i]
        ins = op["in"]["X"]
        if len(ins) != 2 or not getattr(self, "use_patch", True):
            return False
        if not any(self.ops[self.producer[n]]["type"] == "nearest_interp_v2" for n in ins if n in self.producer):
            return False
        cons = self._live_consumers(op["out"]["Out"][0])
        if len(cons) != 1 or self.ops[cons[0]]["type"] != "conv2d":
            return False
        c = self.ops[cons[0]]
        w = self.W[c["in"]["Filter"][0]]
        return list(c["attrs"]["strides"]) == [1, 1] and w.shape[2] * w.shape[3] >= 5 and c["attrs"].get("groups", 1) == 1

    def is_param(self, name):
        return name in self.W

    def new_buf(self, n, h, w, ld, esize=2, ext=None):
        b = Buf(len(self.bufs), n, h, w, ld, esize, ext)
        self.bufs.append(b)
        return b

    def _concat_buf(self, cname, n, h, w):
        """Create the concat buffer once all input channel counts are known (they are: static shapes)."""
        lay = self.concat_layout.get(cname)
        if lay is not None:
            return lay
        op = self.ops[self.producer[cname]]
        offs = []
        tot = 0
        for name in op["in"]["X"]:
            c = self._static_channels(name)
            offs.append((tot, c))
            tot += rup(c, 8)
        buf = self.new_buf(n, h, w, tot)
        lay = {"buf": buf, "offs": offs, "inputs": list(op["in"]["X"])}
        self.concat_layout[cname] = lay
        return lay

    def _static_channels(self, name):
        """Logical channel count of a tensor from the descriptor's var shapes (NCHW dim 1)."""
        if name in self.env:
            return self.env[name].c
        dims = self.desc["var_shapes"].get(name)
        assert dims is not None and len(dims) == 4 and dims[1] > 0, (name, dims)
        return dims[1]

    def wants_lo(self, name):
        """The tensor feeds an OP_CHAIN (chains.py): store it as an fp16 hi + lo pair so that the chain computes on ~22 bits of it
        instead of 11 — the one rounding per chain edge that is left once the intermediates live in LDS.  Ordinary consumers read
        the hi half (a plain fp16 tensor with a wider pixel stride)."""
        if not (getattr(self, "chain", False) and getattr(self, "chain_lo", True)) or self.ragged:
            return False
        if name in self.placement or name in self.fetched_names:
            return False
        for j in self._live_consumers(name):
            o = self.ops[j]
            if o["type"] == "conv2d_transpose" and o["in"]["Input"][0] == name and tuple(self.W[o["in"]["Filter"][0]].shape[2:]) == (2, 2):
                return True          # the DB head's tail runs as a chain (chains.py try_lower_head_tail): right in front of the logits
            if o["type"] not in ("conv2d", "depthwise_conv2d") or o["in"]["Input"][0] != name:
                continue
            if self._chain_candidate(j) is not None or self._is_dw(o, self.W[o["in"]["Filter"][0]].shape[0]):
                return True          # (a depthwise conv filters both halves of a pair: simple_ops.hip)
            wj = self.W[o["in"]["Filter"][0]]
            a = o["attrs"]
            if (o["type"] == "conv2d" and a.get("groups", 1) == 1 and tuple(wj.shape[2:]) == (1, 1) and list(a["strides"]) == [1, 1]
                    and not any(a["paddings"])):
                return True          # a plain 1x1 conv reads both halves as input channels (lower_conv, pair_in)
        return False

    def alloc_out(self, name, n, h, w, c, esize=2, lo=False):
        """Output view for tensor `name`; lands inside a concat buffer slice when planned so.  lo: room for the lo half of an
        fp16 hi + lo pair behind the hi channels of every pixel (Buf.lo_off)."""
        if lo and esize == 2 and self.placement.get(name) is None and PAIR_MIN_PIX <= h * w <= PAIR_MAX_PIX:
            span = rup(c, 8)
            b = self.new_buf(n, h, w, 2 * span, esize)
            b.lo_off = span
            return View(b, 0, n, h, w, [(0, c)], span)
        cname = self.placement.get(name)
        if cname is not None and esize == 2:
            lay = self._concat_buf(cname, n, h, w)
            b = lay["buf"]
            if (b.n, b.h, b.w) == (n, h, w):
                k = lay["inputs"].index(name)
                off, cc = lay["offs"][k]
                assert cc == c, (name, cc, c)
                return View(b, off, n, h, w, [(0, c)], rup(c, 8))
        span = rup(c, 8)
        b = self.new_buf(n, h, w, span, esize)
        return View(b, 0, n, h, w, [(0, c)], span)

    def vrec(self, v: Optional[View]):
        r = ir.empty_view()
        if v is None:
            return r
        b = v.buf
        r["off"] = v.coff * b.esize        # arena base added after allocation
        r["arena"] = -1 - b.id             # patched after allocation
        r["n"], r["h"], r["w"] = v.n, v.src_h, v.src_w
        r["c"], r["ld"], r["esize"] = v.span, b.ld, b.esize
        return r

    # -------------------------------------------------------------------------------------------- exact detector input
    def fold_input_norm(self):
        """The reference feeds the detector (u/255 - mean)/std in fp32 (paddleocr NormalizeImage); rounding
        that to fp16 alone moves the real detector's map by up to 4e-3.  The resized pixels u are integers <= 255, exact in
        fp16: feed THEM (vse_det_preprocess raw mode: channels 0..2 = u, channel 3 = 1 inside the image) and let every conv
        that reads the feed carry the normalisation in its weights:
            w'[co, c, tap] = w[co, c, tap] / (255 std_c)   (c < 3),      w'[co, 3, tap] = - sum_c w[co, c, tap] mean_c / std_c.
        Zero padding of the 4-channel raw input is then the reference's zero padding of the normalised image, exactly (the ones
        channel is 0 outside the image too).  Returns the feed's logical channel count (4)."""
        mean, std = (np.asarray(v, np.float64).reshape(3) for v in self.input_norm)
        feeds = {op["out"]["Out"][0] for op in self.ops if op["type"] == "feed"}
        for k, op in enumerate(self.ops):
            if not any(nm in feeds for names in op["in"].values() for nm in names):
                continue
            if op["type"] != "conv2d" or op["attrs"].get("groups", 1) != 1 or op["in"]["Input"][0] not in feeds:
                raise UnsupportedGraph(f"input_norm: the feed is read by a {op['type']} (only a dense conv can carry the normalisation)")
            wname = op["in"]["Filter"][0]
            w = self.W[wname].astype(np.float64)
            if w.shape[1] != 3:
                raise UnsupportedGraph("input_norm: the stem conv does not take 3 input channels")
            aug = np.zeros((w.shape[0], 4) + w.shape[2:], np.float64)
            aug[:, :3] = w / (255.0 * std).reshape(1, 3, 1, 1)
            aug[:, 3] = -(w * (mean / std).reshape(1, 3, 1, 1)).sum(1)
            new = wname + ":input_norm"
            self.W[new] = aug                                   # fp64: BN folding and the hi + lo split work on it
            self.ops[k] = dict(op, **{"in": dict(op["in"], Filter=[new])})      # (the descriptor's own op record stays untouched)
        return 4

    # -------------------------------------------------------------------------------------------- ragged widths
    def wl_after(self, lvl, k, s, p, ceil=False):
        """Width level behind a k-wide window (stride s, padding p) over level `lvl`."""
        if lvl is None:
            return None
        if s == 1 and k - 1 == 2 * p:
            return lvl
        key = (lvl, k, s, p, bool(ceil))
        if key not in self.wlevel_index:
            self.wlevel_index[key] = len(self.wlevels)
            self.wlevels.append(key)
        return self.wlevel_index[key]

    def sel_width(self, lvl, actual):
        """Width the kernel SELECTION sees for a tensor: in a ragged plan the width of level `lvl` for a nominal sample
        (sel_w0 wide), so that every plan of the model — whatever its batch and widest sample — sends a layer to the same
        kernel family and therefore sums its products in the same order (results are then bit-identical across batch
        compositions); otherwise the tensor's own width."""
        if not self.ragged or lvl is None:
            return actual
        def width(l):
            if l == 0:
                return self.sel_w0
            parent, k, s, p, ceil = self.wlevels[l]
            return int(level_width(width(parent), k, s, p, ceil))
        return width(lvl)

    def _ragged_levels(self, kind, name, ins, out, out2, flags, p):
        """Levels of in0 / of the output of one emitted op (ragged plans); refuses what the masked kernels do not cover."""
        in0 = next((v for v in ins if v is not None), None)
        lin = in0.buf.wl if in0 is not None else None
        for v in list(ins) + [out, out2]:
            if v is not None and (v.up or v.parts is not None):
                raise UnsupportedGraph(f"ragged plan: {name} reads or writes a virtually upsampled / two-source view")
        if kind in (ir.OP_CONV, ir.OP_DWCONV):
            if flags & (ir.F_PIXSHUF | ir.F_DOT1 | ir.F_SRC2 | ir.F_UP2HEAD) or p.get(ir.P_INSHIFT, 0) or p.get(ir.P_RESSHIFT, 0):
                raise UnsupportedGraph(f"ragged plan: {name} uses a conv form without a per-sample width (transposed / fused head / upsampled)")
            lout = self.wl_after(lin, p[ir.P_KW], p[ir.P_SW], p[ir.P_PW])
        elif kind == ir.OP_POOL:
            lout = self.wl_after(lin, p[ir.P_KW], p[ir.P_SW], p[ir.P_PW], bool(p.get(ir.P_POOL_CEIL, 0)))
        elif kind in (ir.OP_GAP, ir.OP_WSCALE):
            lout = None
        elif kind == ir.OP_BINARY:
            if p.get(ir.P_BIN_SHIFT, 0):
                raise UnsupportedGraph(f"ragged plan: {name} adds an upsampled tensor")
            # binary_kernel has no mask of its own: zeros right of a sample come from BOTH operands being masked at the same
            # level and from an activation with f(0) = 0 — anything else would leak values into the padding, silently
            for v in ins:
                if v is not None and v.buf.wl != lin:
                    raise UnsupportedGraph(f"ragged plan: {name} combines tensors of width levels {lin} and {v.buf.wl}")
            if p.get(ir.P_BIN_ACT, 0) in (ir.ACT_SIGMOID, ir.ACT_HSIGMOID):
                raise UnsupportedGraph(f"ragged plan: {name} applies an activation with f(0) != 0 behind an element-wise op")
            lout = lin
        elif kind == ir.OP_RESIZE:
            if p.get(0, 0):
                raise UnsupportedGraph(f"ragged plan: {name} upsamples")
            lout = lin
        else:
            lout = lin
        for v in (out, out2):
            if v is None:
                continue
            if v.buf.wl is not None and v.buf.last >= 0 and v.buf.wl != lout:
                raise UnsupportedGraph(f"ragged plan: {name} writes level {lout} into a buffer of level {v.buf.wl}")
            v.buf.wl = lout
        return lin, lout

    def emit(self, kind, name, ins, out, flags=0, p=None, f=None, w_off=0, b_off=0, aux_off=0, out2=None):
        idx = len(self.ir_ops)
        rec = dict(kind=kind, name=name, flags=flags, p=dict(p or {}), f=dict(f or {}), ins=list(ins), out=out,
                   out2=out2, w_off=w_off, b_off=b_off, aux_off=aux_off)
        if self.ragged:
            lin, lout = self._ragged_levels(kind, name, ins, out, out2, flags, rec["p"])
            rec["p"][ir.P_WLIN] = 0 if lin is None else lin + 1
            rec["p"][ir.P_WLOUT] = 0 if lout is None else lout + 1
        for v in ins:
            if v is not None and v.dense1 and not (kind == ir.OP_CONV and flags & ir.F_UP2HEAD and v is ins[0]):
                raise Tail2Unsupported(f"{name} reads the dense 1-channel map of an F_TAIL2 conv")
        for v in list(ins) + [out, out2]:
            if v is not None and v.buf is not None:
                v.buf.first = min(v.buf.first, idx)
                v.buf.last = max(v.buf.last, idx)
        rec["gmac"] = 0.0
        self.ir_ops.append(rec)
        return rec

    def add_gmacs(self, g):
        """Algorithmic (unpadded) MACs of the op emitted last, in units of 1e9."""
        self.gmacs += g
        self.ir_ops[-1]["gmac"] += g

    def add_weights(self, key, arr):
        """Append to the weight blob (256-byte aligned); identical keys are shared across plans."""
        return self.store.add(key, arr)

    def need_dense(self, v: View, what):
        """Consumers that address channels linearly (depthwise conv, pooling, element-wise ops) need one dense segment."""
        if v.parts is not None:
            v = self.materialize(v, what)
        if v.segs != [(0, v.c)]:
            raise UnsupportedGraph(f"{what}: input channels are laid out in segments {v.segs} (a concat of parts that are not "
                                   "multiples of 8 channels); only convolutions can read such a tensor")
        return v

    # -------------------------------------------------------------------------------------------- materialize
    def materialize(self, v: View, name="mat"):
        """Turn a virtual (upsampled / 2-source) view into a real buffer."""
        if v.parts is not None:
            b = self.new_buf(v.n, v.h, v.w, v.span)
            off = 0
            for pv in v.parts:
                dst = View(b, off, v.n, v.h, v.w, [(0, pv.c)], pv.span)
                self.emit(ir.OP_RESIZE, name + ":part", [pv], dst, p={0: pv.up})
                off += pv.span
            return View(b, 0, v.n, v.h, v.w, list(v.segs), v.span)
        if v.up == 0:
            return v
        out = self.alloc_out("__mat_" + name, v.n, v.h, v.w, v.c)
        if v.segs != [(0, v.c)]:
            out = View(self.new_buf(v.n, v.h, v.w, v.span), 0, v.n, v.h, v.w, list(v.segs), v.span)
        self.emit(ir.OP_RESIZE, name, [v], out, p={0: v.up})
        return out

    # -------------------------------------------------------------------------------------------- resolve
    def resolve(self, name) -> Optional[View]:
        """View for a tensor, looking through not-yet-visited layout-only producers."""
        if name in self.env:
            return self.env[name]
        i = self.producer.get(name)
        if i is None:
            return None
        op = self.ops[i]
        if op["type"] in _VIRTUAL or (op["type"] == "scale" and self._scale_is_noop(op)):
            if self._lower_virtual(i, dry=True):
                return self.env.get(name)
        return None

    @staticmethod
    def _scale_is_noop(op):
        a = op["attrs"]
        return abs(a.get("scale", 1.0) - 1.0) < 1e-12 and abs(a.get("bias", 0.0)) < 1e-12

    def _lower_virtual(self, i, dry=False):
        op = self.ops[i]
        t = op["type"]
        a = op["attrs"]
        if t in ("shape", "fill_constant", "fill_constant_batch_size_like"):
            self.done.add(i)
            return True
        inname = op["in"]["X"][0] if "X" in op["in"] else op["in"]["Input"][0]
        src = self.resolve(inname)
        if src is None:
            return False
        outname = op["out"]["Out"][0]
        v = View(src.buf, src.coff, src.n, src.h, src.w, list(src.segs), src.span, src.up, src.tag, gate=src.gate)
        if t == "nearest_interp_v2":
            s = a["scale"]
            assert s[0] == s[1] and s[0] in (2.0, 4.0, 8.0), s
            sh = int(round(math.log2(s[0])))
            v.up = src.up + sh
            v.h, v.w = src.h << sh, src.w << sh
        elif t == "flatten_contiguous_range":
            assert src.tag == "nchw" and src.h == 1 and a["start_axis"] == 2
            v.tag = "bct"
        elif t == "squeeze2":
            assert src.tag == "nchw" and src.h == 1 and a["axes"] == [2]
            v.tag = "bct"
        elif t == "transpose2":
            perm = tuple(a["axis"])
            trans = {("bct", (0, 2, 1)): "btc", ("btc", (0, 2, 1)): "bct", ("bct", (2, 0, 1)): "tbc",
                     ("tbc", (1, 0, 2)): "btc", ("btc", (1, 0, 2)): "tbc", ("b1tc", (0, 3, 1, 2)): "nchw"}
            key = (src.tag, perm)
            if key not in trans:
                raise NotImplementedError(f"transpose2 {key} at op {i}")
            v.tag = trans[key]
        elif t == "reshape2":
            # only the SVTR tail reshape [B,T,C] -> [B,1,T,C] reaches here (attention reshapes are matched
            # inside _lower_attention)
            assert src.tag == "btc", (src.tag, i)
            v.tag = "b1tc"
        elif t in ("dropout", "assign", "scale"):
            pass
        else:
            raise NotImplementedError(t)
        self.env[outname] = v
        self.done.add(i)
        return True

    # -------------------------------------------------------------------------------------------- epilogue chain
    def _scalar_param(self, name):
        if self.is_param(name) and self.W[name].size == 1:
            return float(self.W[name].reshape(-1)[0])
        return None

    def absorb_epilogue(self, start_name, i_prod, cout, allow_res=True, out_dims=None):
        """Follow the single-consumer chain after a linear op and fold what the kernel epilogue can do.

        Returns dict(scale[c], shift[c], act, act_a, act_b, post_a, post_b, res(View|None), res_first, act2,
                     out_name)."""
        scale = np.ones(cout, np.float64)
        shift = np.zeros(cout, np.float64)
        st = dict(act=ir.ACT_NONE, act_a=0.0, act_b=0.0, post_a=1.0, post_b=0.0, res=None, act2=ir.ACT_NONE)
        name = start_name
        stage = 0   # 0: pre-act affine, 1: post-act scalar affine, 2: after residual, 3: closed
        while True:
            cons = self._live_consumers(name)
            if len(cons) != 1 or name in self.placement:
                # a tensor that is a concat input may still be followed by fusable ops only if that consumer is
                # the concat itself -> stop
                break
            j = cons[0]
            op = self.ops[j]
            t = op["type"]
            a = op["attrs"]
            if t in ("elementwise_add", "elementwise_mul"):
                x, y = op["in"]["X"][0], op["in"]["Y"][0]
                other = y if x == name else x
                if self.is_param(other):
                    pv = self.W[other].astype(np.float64).reshape(-1)
                    if pv.size == 1:
                        s = float(pv[0])
                        if stage == 0:
                            if t == "elementwise_mul":
                                scale *= s
                                shift *= s
                            else:
                                shift += s
                        elif stage == 1:
                            if t == "elementwise_mul":
                                st["post_a"] *= s
                                st["post_b"] *= s
                            else:
                                st["post_b"] += s
                        else:
                            break
                    elif pv.size == cout and stage == 0:
                        if t == "elementwise_mul":
                            scale *= pv
                            shift *= pv
                        else:
                            shift += pv
                    else:
                        break
                elif (t == "elementwise_add" and allow_res and stage <= 1 and st["res"] is None
                      and getattr(self, "chain_res", None) is not None and other in self.chain_res):
                    st["res"] = ("chain", other)          # chains.py: the residual is a channel-minor LDS buffer of the same chain
                    stage = 2
                elif t == "elementwise_add" and allow_res and stage <= 1 and st["res"] is None:
                    rv = self.resolve(other)
                    if rv is None or rv.tag not in ("nchw", "btc", "tbc") or rv.segs != [(0, cout)]:
                        break
                    if out_dims is not None and (rv.n, rv.h, rv.w) != tuple(out_dims):
                        break
                    st["res"] = rv
                    stage = 2
                else:
                    break
            elif t == "batch_norm" and stage == 0:
                g = self.W[op["in"]["Scale"][0]].astype(np.float64)
                b = self.W[op["in"]["Bias"][0]].astype(np.float64)
                m = self.W[op["in"]["Mean"][0]].astype(np.float64)
                var = self.W[op["in"]["Variance"][0]].astype(np.float64)
                k = g / np.sqrt(var + a["epsilon"])
                scale = scale * k
                shift = (shift - m) * k + b
                name = op["out"]["Y"][0]
                self.done.add(j)
                continue
            elif t in _ACTS:
                if stage == 0:
                    st["act"] = _ACTS[t]
                    if t == "hard_sigmoid":
                        st["act_a"], st["act_b"] = a["slope"], a["offset"]
                    stage = 1
                elif stage == 2 and t == "relu":
                    st["act2"] = ir.ACT_RELU
                    stage = 3
                else:
                    break
            elif t in ("dropout", "assign") or (t == "scale" and self._scale_is_noop(op)):
                pass
            else:
                break
            name = op["out"]["Out"][0]
            self.done.add(j)
        st.update(scale=scale, shift=shift, out_name=name)
        return st

    # -------------------------------------------------------------------------------------------- conv lowering
    def pack_conv_weights(self, w, scale, inv: View, pixshuf=False):
        """w: [Cout,Cin,kh,kw] fp32 (already transposed for convT).  Returns (matrix [Np][Kp], Np, Kp)."""
        cout, cin, kh, kw = w.shape
        w = w.astype(np.float64) * scale.reshape(-1, 1, 1, 1)
        cinp = inv.span
        coutp = rup(cout, 8)
        full = np.zeros((coutp, kh, kw, cinp), np.float64)
        full[:cout][:, :, :, inv.chmap()] = np.transpose(w, (0, 2, 3, 1))
        K = kh * kw * cinp
        Kp = rup(K, ir.KT)
        mat = np.zeros((coutp, Kp), np.float64)
        mat[:, :K] = full.reshape(coutp, K)
        return mat, coutp, Kp

    @staticmethod
    def pw_weights(mat, hilo=False):
        """conv_pw_kernel: plain [Np][cinp] fp16; hilo: the table of lo = fp16(w - hi) follows the table of hi = fp16(w)."""
        hi = np.asarray(mat, np.float64).astype(np.float16)
        if not hilo:
            return hi.reshape(-1)
        return np.concatenate([hi.reshape(-1), (np.asarray(mat, np.float64) - hi.astype(np.float64)).astype(np.float16).reshape(-1)])

    @staticmethod
    def tile_weights(mat, kt=ir.KT, hilo=False):
        """[Np][Kp] -> [Kp/kt][Np][kt] fp16; hilo: the tiles of hi = fp16(w) followed by the tiles of lo = fp16(w - hi)."""
        npad, kp = mat.shape
        t = np.ascontiguousarray(mat.reshape(npad, kp // kt, kt).transpose(1, 0, 2))
        hi = t.astype(np.float16)
        if not hilo:
            return hi
        lo = (t - hi.astype(np.float64)).astype(np.float16)
        return np.concatenate([hi.reshape(-1), lo.reshape(-1)])

    @staticmethod
    def stem_weights(mat, hilo=False):
        """[Np][9 taps x 8 padded channels (+ K padding)] -> [Np][10 taps][8 channels] fp16 (tap 9 and channels 4..7 zero):
        the generic kernels' K order, so the accumulation order (and every output bit) stays theirs."""
        npad = mat.shape[0]
        full = mat[:, :72].reshape(npad, 9, 8)
        assert not full[:, :, 4:].any(), "stem kernel: more than 4 real input channels"
        out = np.zeros((npad, 10, 8), np.float64)
        out[:, :9] = full
        hi = out.reshape(-1).astype(np.float16)
        if not hilo:
            return hi
        return np.concatenate([hi, (out.reshape(-1) - hi.astype(np.float64)).astype(np.float16)])

    def _is_dw(self, o2, c):
        if o2["type"] not in ("conv2d", "depthwise_conv2d"):
            return False
        w2 = self.W[o2["in"]["Filter"][0]]
        g2 = o2["attrs"].get("groups", 1)
        return (o2["type"] == "depthwise_conv2d" or (g2 > 1 and g2 == w2.shape[0])) and w2.shape[1] == 1 and w2.shape[0] == c

    def _gate_foldable(self, name, big, flags):
        """The SE output `name` (= big * gate) is read only by consumers that can apply the gate themselves: at most one
        depthwise conv and any number of plain 1x1 stride-1 convs that run on conv_gemm_kernel over the whole dense tensor."""
        if flags or name in self.placement or big.up or big.segs != [(0, big.c)] or big.span != big.c or big.c % 64 or big.c <= 64 or self.hilo:
            return False
        if name in self.fetched_names:       # a fetched SE output must exist as a tensor: env[name] = big would hand out the un-gated one
            return False
        cons = self._live_consumers(name)
        n_dw = n_pw = 0
        for j in cons:
            o2 = self.ops[j]
            if o2["type"] not in ("conv2d", "depthwise_conv2d") or o2["in"]["Input"][0] != name:
                return False
            if self._is_dw(o2, big.c):
                n_dw += 1
                continue
            a2 = o2["attrs"]
            w2 = self.W[o2["in"]["Filter"][0]]
            pads = a2["paddings"]
            if (o2["type"] != "conv2d" or a2.get("groups", 1) != 1 or tuple(w2.shape[1:]) != (big.c, 1, 1) or list(a2["strides"]) != [1, 1]
                    or any(pads) or big.h * self.sel_width(big.buf.wl if big.buf is not None else None, big.w) < 256):
                return False
            n_pw += 1
        return n_dw <= 1 and n_pw >= 1

    @staticmethod
    def gemm_eligible(kh, kw, ph, pw, cinp, inshift, flags):
        """Mirror of conv_gemm_mode() (csrc/conv_gemm.hip): the layer runs on conv_gemm_kernel.  The launcher refuses an
        F_WK32 op it cannot send there, so a drift between the two rules fails loudly instead of computing garbage."""
        if _dev_switch("VSE_CONV_GEMM", "1")[:1] == "0":
            return False
        if inshift or (flags & (ir.F_PATCH | ir.F_DOT1 | ir.F_SRC2 | ir.F_UP2HEAD)):
            return False
        return cinp % 32 == 0 and kh * kw <= 31 and kh >= 2 * ph + 1 and kw >= 2 * pw

    def _dot1_candidate(self, name, cout):
        """Structural half of _try_fuse_dot1 without side effects: the only consumer is a 1x1 conv to ONE channel."""
        cons = self._live_consumers(name)
        if len(cons) != 1 or name in self.placement or self.ops[cons[0]]["type"] != "conv2d":
            return False
        op = self.ops[cons[0]]
        a = op["attrs"]
        return (tuple(self.W[op["in"]["Filter"][0]].shape) == (1, cout, 1, 1) and list(a["strides"]) == [1, 1]
                and not any(a["paddings"]) and a.get("groups", 1) == 1)

    def _try_fuse_dot1(self, name, cout, coutp):
        """`name` (conv output after its epilogue) -> conv2d 1x1 to ONE channel (+bias, +sigmoid): fold it into the
        producing conv's epilogue as a per-pixel dot product; the wide tensor is then never written to HBM."""
        cons = self._live_consumers(name)
        if len(cons) != 1 or name in self.placement:
            return None
        j = cons[0]
        op = self.ops[j]
        if op["type"] != "conv2d":
            return None
        a = op["attrs"]
        w2 = self.W[op["in"]["Filter"][0]]
        if tuple(w2.shape) != (1, cout, 1, 1) or list(a["strides"]) != [1, 1] or any(a["paddings"]) or a.get("groups", 1) != 1:
            return None
        done_before = set(self.done)
        ep2 = self.absorb_epilogue(op["out"]["Output"][0], j, 1, allow_res=False)
        ok = (ep2["act"] in (ir.ACT_NONE, ir.ACT_SIGMOID) and ep2["post_a"] == 1.0 and ep2["post_b"] == 0.0
              and ep2["act2"] == ir.ACT_NONE)
        if not ok:
            self.done = done_before
            return None
        self.done.add(j)
        wv = np.zeros(coutp, np.float32)
        wv[:cout] = w2[0, :, 0, 0].astype(np.float64) * ep2["scale"][0]
        out_name = ep2["out_name"]
        x = self.env_dims_tmp
        n, h, w = x
        fcons = self._live_consumers(out_name)
        if len(fcons) == 1 and self.ops[fcons[0]]["type"] == "fetch":
            ob = self.new_buf(n, h, w, 1, esize=4, ext=len(self.outputs) + 1)
            self.outputs.append(dict(name=out_name, kind="map", n=n, h=h, w=w, c=1, ld=1, esize=4))
        else:
            ob = self.new_buf(n, h, w, 8, esize=2)
        view = View(ob, 0, n, h, w, [(0, 1)], 1 if ob.esize == 4 else 8)
        return dict(w=wv, b=float(ep2["shift"][0]), act=ep2["act"], out_name=out_name, view=view,
                    wname=op["in"]["Filter"][0])

    @staticmethod
    def patch_weights(mat, kh, kw, cinp, tp):
        """[Np][Kp] (K order tap-major (dy, dx), channel-minor) -> [ceil(cinp/32)][tp taps][Np][32] fp16 for
        conv_patch_kernel, taps in COLUMN-major order (t' = dx*kh + dy: the kernel walks filter columns so that
        consecutive taps share an activation fragment); zero weights for the channel tail and for the taps appended
        up to `tp` = whole kernel steps."""
        npad = mat.shape[0]
        c32 = rup(cinp, 32)
        taps = kh * kw
        col_major = [(t % kh) * kw + t // kh for t in range(taps)]        # stream position t' -> row-major tap
        full = np.zeros((npad, tp, c32), mat.dtype)
        full[:, :taps, :cinp] = mat[:, :taps * cinp].reshape(npad, taps, cinp)[:, col_major, :]
        m = np.ascontiguousarray(full.reshape(npad, tp, c32 // 32, 32).transpose(2, 1, 0, 3)).astype(np.float16)
        # PATCH_WPAD_STEPS = 4 zero steps (of up to 4 taps) after the stream: the kernel's DMA look-ahead runs past the
        # last real step without a bounds test and must land on readable zeros
        return np.concatenate([m.reshape(-1), np.zeros(16 * npad * 32, np.float16)])

    @staticmethod
    def col_weights(mat, kh, kw, cinp, hilo=False):
        """[Np][Kp] (K order tap-major (dy, dx), channel-minor) -> [cinp/16][kw][kh][Np][16] fp16 for conv_col_kernel /
        conv_c3_kernel: one ring stage = one filter column of one 16-channel chunk, contiguous; hilo: the stream of
        lo = fp16(w - hi) follows the stream of hi = fp16(w) (second pass over the same chunks); three zero stages follow (the
        DMA look-ahead of the last steps lands on readable zeros)."""
        npad = mat.shape[0]
        assert cinp % 16 == 0
        full = mat[:, :kh * kw * cinp].reshape(npad, kh, kw, cinp // 16, 16)
        m = np.ascontiguousarray(full.transpose(3, 2, 1, 0, 4))
        hi = m.astype(np.float16)
        parts = [hi.reshape(-1)]
        if hilo:
            parts.append((m - hi.astype(np.float64)).astype(np.float16).reshape(-1))
        return np.concatenate(parts + [np.zeros(3 * kh * npad * 16 + 512, np.float16)])

    @staticmethod
    def head_up2_weights(mat, cinp):
        """3x3 conv over concat[u (8 physical channels, 1 real), up2(x) (64 channels)], matrix [Np][9*cinp] in
        (tap, channel) order -> the stream of conv_head_up2_kernel:
            [chunk 0..1][parity a*2+b][tap r*2+s][64][32] fp16   (x part, folded onto the low-res grid)
            [64][32] fp16                                        (u part: k = 3*dy + dx < 9, rest zero)
        Under nearest x2 upsampling output parity a sees low-res row offsets a-1 (r=0) and a (r=1); the original taps
        that land on the same low-res pixel are summed (fp64): a=0: r0 <- dy 0, r1 <- dy 1,2;  a=1: r0 <- dy 0,1,
        r1 <- dy 2 (same for columns)."""
        npad = mat.shape[0]
        assert npad <= 64 and cinp == 72
        w = np.zeros((64, 3, 3, cinp), np.float64)
        w[:npad] = mat[:, :9 * cinp].reshape(npad, 3, 3, cinp)
        wx, wu = w[..., 8:72], w[..., 0]                       # [64,3,3,64], [64,3,3]
        taps_of = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}
        out = np.zeros((2, 4, 4, 64, 32), np.float64)
        for a in range(2):
            for b in range(2):
                for r in range(2):
                    for s_ in range(2):
                        acc = np.zeros((64, 64), np.float64)
                        for dy in taps_of[(a, r)]:
                            for dx in taps_of[(b, s_)]:
                                acc += wx[:, dy, dx, :]
                        for c in range(2):
                            out[c, a * 2 + b, r * 2 + s_] = acc[:, c * 32:(c + 1) * 32]
        ublock = np.zeros((64, 32), np.float64)
        ublock[:, :9] = wu.reshape(64, 9)
        return np.concatenate([out.reshape(-1), ublock.reshape(-1)]).astype(np.float16)

    def _tail2_candidate(self, name, c1):
        """`name` = the output (behind its BN / activation) of a 2x2 s2 transposed conv with c1 couts.  If ONE of its readers is a second
        2x2 s2 transposed conv c1 -> 1 whose own output (behind its activation) is read only as the FIRST part of a virtual concat — the
        PP-OCRv4 server detector's base map in front of the local refinement conv — claim that conv and its epilogue and return
        {ep, w, wname}; else None with nothing changed."""
        cons = [k for k in self._live_consumers(name) if self.ops[k]["type"] == "conv2d_transpose" and self.ops[k]["in"]["Input"][0] == name
                and k not in self.done]
        if len(cons) != 1 or name in self.fetched_names:
            return None
        k = cons[0]
        op1 = self.ops[k]
        a1 = op1["attrs"]
        wn = op1["in"]["Filter"][0]
        w2 = self.W[wn]
        if tuple(w2.shape) != (c1, 1, 2, 2) or list(a1["strides"]) != [2, 2] or any(a1["paddings"]) or a1.get("groups", 1) != 1:
            return None
        snapshot = set(self.done)
        ep2 = self.absorb_epilogue(op1["out"]["Output"][0], k, 1, allow_res=False)
        fc = self._live_consumers(ep2["out_name"])
        ok = (ep2["post_a"] == 1.0 and ep2["post_b"] == 0.0 and ep2["act2"] == ir.ACT_NONE and ep2["act"] in (ir.ACT_NONE, ir.ACT_SIGMOID, ir.ACT_RELU)
              and ep2["out_name"] not in self.fetched_names and ep2["out_name"] not in self.placement and len(fc) == 1
              and self.ops[fc[0]]["type"] == "concat" and self.ops[fc[0]]["out"]["Out"][0] in self.virtual_concats
              and self.ops[fc[0]]["in"]["X"][0] == ep2["out_name"] and len(self.ops[fc[0]]["in"]["X"]) == 2)
        if not ok:
            self.done = snapshot
            return None
        self.done.add(k)
        return dict(ep=ep2, w=w2, wname=wn)

    def lower_conv(self, i):
        op = self.ops[i]
        a = op["attrs"]
        t = op["type"]
        wname = op["in"]["Filter"][0]
        w = self.W[wname]
        inv = self.resolve(op["in"]["Input"][0])
        assert inv is not None and inv.tag == "nchw", (i, op["in"])
        outname = op["out"]["Output"][0]
        sh, sw = a["strides"]
        pads = a["paddings"]
        ph, pw = (pads[0], pads[1]) if len(pads) == 2 else (pads[0], pads[2])
        groups = a.get("groups", 1)
        if t == "depthwise_conv2d" or (groups > 1 and groups == w.shape[0] and w.shape[1] == 1):
            return self.lower_dwconv(i, inv, w, sh, sw, ph, pw)
        assert groups == 1, "only dense and depthwise convs occur (SURVEY App. E)"
        if t == "conv2d_transpose":
            assert (sh, sw) == (2, 2) and w.shape[2:] == (2, 2) and (ph, pw) == (0, 0)
            cin, cout = w.shape[0], w.shape[1]
            ep = self.absorb_epilogue(outname, i, cout, allow_res=False)
            inv = self.materialize(inv, outname)
            coutp = rup(cout, 8)
            # GEMM N ordered (dy,dx,co): W2[(dy,dx,co), ci]
            w2 = np.zeros((4 * coutp, inv.span), np.float64)
            wt = w.astype(np.float64) * ep["scale"].reshape(1, -1, 1, 1)
            cm = inv.chmap()
            for dy in range(2):
                for dx in range(2):
                    blk = np.zeros((coutp, inv.span))
                    blk[:cout][:, cm] = wt[:, :, dy, dx].T
                    w2[(dy * 2 + dx) * coutp:(dy * 2 + dx + 1) * coutp] = blk
            Kp = rup(inv.span, ir.KT)
            mat = np.zeros((4 * coutp, Kp))
            mat[:, :inv.span] = w2
            bias = np.zeros(4 * coutp, np.float32)
            for q in range(4):
                bias[q * coutp:q * coutp + cout] = ep["shift"]
            oh, ow = inv.h * 2, inv.w * 2
            tflags = ir.F_PIXSHUF | (ir.F_HILO if self.hilo else 0)
            cons = self._live_consumers(ep["out_name"])
            if cout == 1 and len(cons) == 1 and self.ops[cons[0]]["type"] == "fetch" and ONECH:
                # the DB head's last layer (transposed conv to ONE channel + sigmoid) feeds the fetch: store the fp32 map itself
                # (one float per output pixel) instead of 8-channel fp16 groups that a copy pass then narrows: 4x fewer bytes
                # written and no copy (F_ONECH)
                ob = self.new_buf(inv.n, oh, ow, 1, esize=4, ext=len(self.outputs) + 1)
                self.outputs.append(dict(name=ep["out_name"], kind="map", n=inv.n, h=oh, w=ow, c=1, ld=1, esize=4))
                out = View(ob, 0, inv.n, oh, ow, [(0, 1)], 1)
                tflags |= ir.F_OUT_F32 | ir.F_ONECH
            else:
                out = self.alloc_out(ep["out_name"], inv.n, oh, ow, cout)
            pw_ok = PW and inv.span % 8 == 0 and inv.span <= 64 and 4 * coutp <= (128 if self.hilo else 256) and inv.up == 0
            tail = None
            if (pw_ok and TAIL2 and self.tail2 is not False and not self.hilo and not self.ragged and inv.span in (32, 64) and cout > 1
                    and (4 * coutp) % 32 == 0 and not (tflags & ir.F_ONECH) and ep["post_a"] == 1.0 and ep["post_b"] == 0.0
                    and ep["act2"] == ir.ACT_NONE and out.buf.lo_off == 0):
                tail = self._tail2_candidate(ep["out_name"], cout)
            out2, aux_off, tp, tf = None, 0, {}, {}
            if tail is not None:
                # the head's SECOND transposed conv (c1 -> 1, 2x2 s2, + activation) in this launch (F_TAIL2, conv_pw_tail_kernel): stage B
                # = a block-diagonal 1x1 conv over this op's 4 coutp channels (dy, dx, co) -> 16 outputs 4 r + c = pixel (4 y + r, 4 x + c)
                # of the map, r = 2 dy + ey, c = 2 dx + ex; its fp16 weights are the separate launch's (w2 * scale2 rounded once)
                ep2, w2 = tail["ep"], tail["w"].astype(np.float64) * float(tail["ep"]["scale"][0])
                wb = np.zeros((16, 4 * coutp), np.float64)
                for r in range(4):
                    for c in range(4):
                        q = ((r >> 1) * 2 + (c >> 1)) * coutp
                        wb[4 * r + c, q:q + cout] = w2[:, 0, r & 1, c & 1]
                fx = np.arange(16)
                rows = (fx & ~12) | ((fx & 4) << 1) | ((fx & 8) >> 1)           # conv_wrow: the cout row MFMA row fx carries
                nks2 = 4 * coutp // 16
                frag = np.zeros((nks2, 2, 16, 8), np.float16)
                for s_ in range(nks2):
                    for fj in range(2):
                        frag[s_, fj] = wb[rows][:, s_ * 16 + fj * 8:s_ * 16 + fj * 8 + 8].astype(np.float16)
                aux_off = self.add_weights(("convTtail", wname, tail["wname"], ep2["out_name"]), frag.reshape(-1))
                ub = self.new_buf(inv.n, 2 * oh, 2 * ow, 1, esize=2)
                out2 = View(ub, 0, inv.n, 2 * oh, 2 * ow, [(0, 1)], 8, dense1=True)
                tflags |= ir.F_TAIL2
                tp = {ir.P_DOTACT: ep2["act"]}
                tf = {ir.FS_PRE_B: float(np.float32(ep2["shift"][0]))}
                assert ep2["act"] in (ir.ACT_NONE, ir.ACT_SIGMOID, ir.ACT_RELU), ep2      # (activations without parameters: the kernel passes none)
            if pw_ok:
                # few input channels: conv_pw_kernel streams the pixels straight from global memory (pixel-shuffle store as ever);
                # hi + lo weights: two tables, the K slices walked twice over the same activation fragments
                tflags |= ir.F_PW
                Kp = rup(inv.span, 16)
                w_off = self.add_weights(("convTpw", wname, tuple(inv.segs), ep["out_name"], self.hilo), self.pw_weights(mat[:, :Kp], self.hilo))
            else:
                w_off = self.add_weights(("convT", wname, tuple(inv.segs), ep["out_name"], self.hilo),
                                         self.tile_weights(mat, hilo=self.hilo))
            b_off = self.add_weights(("convTb", wname, ep["out_name"]), bias)
            self.emit(ir.OP_CONV, ep["out_name"], [inv], out, flags=tflags,
                      p={ir.P_KH: 1, ir.P_KW: 1, ir.P_SH: 1, ir.P_SW: 1, ir.P_PH: 0, ir.P_PW: 0,
                         ir.P_ACT: ep["act"], ir.P_ACT2: 0, ir.P_COUT: 4 * coutp, ir.P_KTOT: Kp,
                         ir.P_INSHIFT: 0, ir.P_RESSHIFT: 0, ir.P_CINP: inv.span, **tp},
                      f={ir.FS_ACT_A: ep["act_a"], ir.FS_ACT_B: ep["act_b"], ir.FS_POST_A: ep["post_a"],
                         ir.FS_POST_B: ep["post_b"], **tf}, w_off=w_off, b_off=b_off, aux_off=aux_off, out2=out2)
            self.add_gmacs(inv.n * inv.h * inv.w * cin * cout * 4 / 1e9)
            self.env[ep["out_name"]] = out
            if tail is not None:
                self.add_gmacs(inv.n * oh * ow * cout * 4 / 1e9)
                self.env[tail["ep"]["out_name"]] = out2
            return
        cout, cin, kh, kw = w.shape
        assert cin == inv.c, (cin, inv.c, outname)
        pair_in = False
        if (getattr(self, "chain", False) and getattr(self, "chain_lo", True) and (kh, kw, sh, sw, ph, pw) == (1, 1, 1, 1, 0, 0)
                and inv.buf is not None and inv.buf.lo_off and inv.parts is None and inv.up == 0 and inv.coff == 0
                and inv.segs == [(0, inv.c)] and op["in"]["Input"][0] not in self.pending_wgate):
            # the input is an fp16 hi + lo PAIR (it also feeds an OP_CHAIN, or was stored for this conv): a 1x1 conv consumes both
            # halves with NO kernel change — the lo channels are more input channels with the same weights, W hi + W lo
            pair_in = True
            lo_off = inv.buf.lo_off
            inv = View(inv.buf, 0, inv.n, inv.h, inv.w, [(0, inv.c), (lo_off, inv.c)], 2 * lo_off, 0, inv.tag)
            w = np.concatenate([w, w], axis=1)
            cin *= 2
        oh = (inv.h + 2 * ph - kh) // sh + 1
        ow = (inv.w + 2 * pw - kw) // sw + 1
        ep = self.absorb_epilogue(outname, i, cout, out_dims=(inv.n, oh, ow))
        coutp, Kp = rup(cout, 8), rup(kh * kw * inv.span, ir.KT)
        bias = np.zeros(coutp, np.float32)
        bias[:cout] = ep["shift"]
        out = self.alloc_out(ep["out_name"], inv.n, oh, ow, cout, lo=self.wants_lo(ep["out_name"]))
        res = ep["res"]
        flags = 0
        # width the kernel selection sees: the map's own — or, in a ragged plan, the map width of a nominal sample, so that a
        # layer runs on the same kernel family (same summation order) in every plan of the model
        ow_real = ow
        if self.ragged:
            if inv.buf is None:
                raise UnsupportedGraph(f"ragged plan: conv {outname} reads a virtual concat")
            ow = self.sel_width(self.wl_after(inv.buf.wl, kw, sw, pw), ow)
        # k x k stride-1 convs on maps that tile well into 8x32 output patches go to the LDS-resident-patch kernel
        pbn = 64 if rup(coutp, 64) < rup(coutp, 128) else 128          # mirrors conv_patch_bn / conv_patch_th in csrc
        pcap = 960 if pbn == 64 else 640
        th = 16 if (pbn == 64 and (16 + kh - 1) * (32 + kw - 1) <= pcap and -(-oh // 16) * 16 * 100 <= -(-oh // 8) * 8 * 112) else 8
        tile_eff = (oh * ow) / float(-(-oh // th) * th * -(-ow // 32) * 32)
        # (one block per CU: the fixed prologue/epilogue only amortises over a long enough K loop)
        patch_std = ((sh, sw) == (1, 1) and kh * kw >= 5 and (8 + kh - 1) * (32 + kw - 1) <= 640
                     and tile_eff >= PATCH_MIN_TILE_EFF and kh * kw * cin >= PATCH_MIN_K and coutp <= PATCH_MAX_COUT and self.use_patch)
        # LIGHT variant (conv_patch_plan in csrc/conv_patch.hip): 8-row tiles whose halo patch fits 352 pixels (3x3, 1xk),
        # 64 or 128 couts per tile, two blocks per CU; not combined with the fused 1-channel projection or a virtual concat
        tile_eff8 = (oh * ow) / float(-(-oh // 8) * 8 * -(-ow // 32) * 32)
        light_ok = ((sh, sw) == (1, 1) and kh * kw >= 5 and (8 + kh - 1) * (32 + kw - 1) <= 352 and tile_eff8 >= PATCH_MIN_TILE_EFF
                    and kh * kw * cin >= PATCH_MIN_K and self.use_patch and inv.parts is None
                    and (PATCH_LIGHT >= 2 if coutp <= 64 else (coutp <= 128 and PATCH_LIGHT >= 1)))
        # column-per-step kernel (conv_col.hip, mirrors conv_col_ok): tall filters, <= 64 couts, 16-row tiles whose waves
        # below the map idle (a partial tile row costs ~0.35 + 0.65 * live waves / 8 of a full one)
        rem16 = oh % 16
        rows16 = oh // 16 + ((0.35 + 0.65 * -(-rem16 // 2) / 8.0) if rem16 else 0.0)
        tile_eff_col = (oh * ow) / float(rows16 * 16 * -(-ow // 32) * 32)
        col = (COL and (sh, sw) == (1, 1) and kh in (5, 7, 9) and 3 <= kw <= 17 and inv.span % 16 == 0 and coutp <= 64
               and inv.parts is None and self.use_col and kh * kw * cin >= PATCH_MIN_K
               and tile_eff_col >= COL_MIN_TILE_EFF and not self._dot1_candidate(ep["out_name"], cout))
        c3 = (COL3 and (sh, sw) == (1, 1) and (kh, kw, ph, pw) == (3, 3, 1, 1) and inv.span % 16 == 0 and inv.parts is None
              and self.use_col and kh * kw * cin >= min(PATCH_MIN_K, COL3_MIN_K) and coutp <= COL3_MAX_COUT
              and (coutp <= 64 or inv.span >= COL3_WIDE_MIN_CIN)
              and inv.src_h * inv.src_w * inv.buf.ld < 2_000_000_000      # 32-bit in-image offsets (launch_conv_c3 checks the same)
              and c3_tile_eff(oh, ow) >= COL3_MIN_TILE_EFF and not self._dot1_candidate(ep["out_name"], cout))
        col = col or c3
        if col:
            patch_std = light_ok = False
        patch = patch_std or light_ok
        ow = ow_real
        self.env_dims_tmp = (inv.n, oh, ow)
        if patch:
            flags |= ir.F_PATCH
        if col:
            flags |= ir.F_COL
        in2shift = 0
        if inv.parts is not None:
            if patch:
                flags |= ir.F_SRC2
                part0, part1 = inv.parts
                in2shift = part1.up
                inv_main = part0
            else:
                inv = self.materialize(inv, outname)
                inv_main = inv
        else:
            inv_main = inv
        ins = [inv_main]
        resshift = 0
        if res is not None:
            assert (res.n, res.h, res.w, res.c) == (inv.n, oh, ow, cout), (outname, res, oh, ow, cout)
            assert res.segs == [(0, cout)]
            flags |= ir.F_RES
            ins.append(res)
            resshift = res.up
        if flags & ir.F_SRC2:
            while len(ins) < 2:
                ins.append(None)
            ins.append(inv.parts[1])
        # (conv_patch_kernel's fused projection: one cout tile, no residual — launch_conv_patch refuses the rest)
        dot = self._try_fuse_dot1(ep["out_name"], cout, coutp) if (patch_std and th == 16 and coutp <= pbn and res is None) else None
        if patch:
            # taps padded to whole kernel steps (2 taps in the LIGHT variant, else 4), channels to 32
            light = light_ok and dot is None
            assert light or patch_std
            big = (not light) and th == 16 and (16 + kh - 1) * (32 + kw - 1) > 640
            ptaps = rup(kh * kw, 2 if light else 4)
            Kp = ptaps * rup(inv.span, 32)
        # DB head of the PP-OCRv4 server detector: 3x3 over [1-channel full-res map, x2-upsampled 64-channel map] with
        # the fused 1-channel projection -> evaluated on the low-res grid with folded 2x2 taps (conv_head.hip)
        head = (dot is not None and (flags & ir.F_SRC2) and res is None and (kh, kw, ph, pw) == (3, 3, 1, 1)
                and in2shift == 1 and inv_main.up == 0 and inv_main.span == 8 and inv_main.c == 1
                and inv.parts[1].span == 64 and inv.span == 72 and coutp <= 64
                and (oh, ow) == (inv.parts[1].h, inv.parts[1].w) and oh % 2 == 0 and ow % 2 == 0 and HEAD_UP2 and not self.hilo)
        if head:
            flags |= ir.F_UP2HEAD
            Kp = 2 * 4 * 4 * 32 + 32
            w_off = self.add_weights(("convh", wname, tuple(inv.segs), ep["out_name"]),
                                     lambda: self.head_up2_weights(self.pack_conv_weights(w, ep["scale"], inv)[0], inv.span))
        elif (PW and (kh, kw, sh, sw, ph, pw) == (1, 1, 1, 1, 0, 0) and inv.parts is None and inv_main.up == 0 and dot is None
              and inv.span % 8 == 0 and inv.span <= (96 if self.hilo else 64) and coutp <= (128 if self.hilo else PW_MAX_COUT) and flags in (0, ir.F_RES)):
            # (hi + lo nets: the alternative is the generic kernel with K padded to 64 and walked twice — any cout count it can hold
            # is faster here)
            flags |= ir.F_PW | (ir.F_HILO if self.hilo else 0)
            Kp = rup(inv.span, 16)          # weight rows are whole 16-channel K slices (zero columns behind the channels)
            w_off = self.add_weights(("convpw", wname, tuple(inv.segs), ep["out_name"], self.hilo),
                                     lambda: self.pw_weights(self.pack_conv_weights(w, ep["scale"], inv)[0][:, :rup(inv.span, 16)], self.hilo))
        elif col and self.hilo and HLSUM and (kh, kw) == (3, 3) and coutp <= 32:
            # the 32-cout tile of conv_c3_kernel walks K twice for a hi + lo net with half its MFMA tile empty: ONE pass over a 64-row
            # stage [hi 32 | lo 32] instead, the two accumulator tiles added in the epilogue (F_HLSUM)
            Kp = kh * kw * inv.span
            flags |= ir.F_HLSUM

            def pack_hl():
                mat = np.zeros((32, Kp), np.float64)
                m0 = self.pack_conv_weights(w, ep["scale"], inv)[0]
                mat[:m0.shape[0]] = m0[:, :Kp]
                hi = mat.astype(np.float16).astype(np.float64)
                return self.col_weights(np.concatenate([hi, mat - hi]), kh, kw, inv.span, False)
            w_off = self.add_weights(("convc_hl", wname, tuple(inv.segs), ep["out_name"]), pack_hl)
        elif col:
            Kp = kh * kw * inv.span
            if self.hilo:
                flags |= ir.F_HILO
            w_off = self.add_weights(("convc", wname, tuple(inv.segs), ep["out_name"], self.hilo),
                                     lambda: self.col_weights(self.pack_conv_weights(w, ep["scale"], inv)[0], kh, kw, inv.span,
                                                              self.hilo))
        elif patch:
            # (the tap padding depends on the kernel variant the map size selects: part of the cache key)
            w_off = self.add_weights(("convp", wname, tuple(inv.segs), ep["out_name"], ptaps),
                                     lambda: self.patch_weights(self.pack_conv_weights(w, ep["scale"], inv)[0], kh, kw,
                                                                inv.span, ptaps))
        elif (STEM and (kh, kw, ph, pw) == (3, 3, 1, 1) and (sh, sw) in ((1, 1), (2, 2)) and inv.span == 8 and cin <= 4
              and coutp <= 64 and inv.parts is None and inv_main.up == 0 and dot is None and flags in (0, ir.F_RES)):
            # stem over an image-like input (conv_stem.hip)
            flags |= ir.F_STEM | (ir.F_HILO if self.hilo else 0)
            if self.fuse_preprocess and inv_main.buf.ext == 0:
                flags |= ir.F_U8SRC              # the detector's pre-processing rides in the stem's patch staging (conv_stem.hip)
                self._u8_fused = True
            w_off = self.add_weights(("convs", wname, tuple(inv.segs), ep["out_name"], self.hilo),
                                     lambda: self.stem_weights(self.pack_conv_weights(w, ep["scale"], inv)[0], self.hilo))
        else:
            wk32 = WK32 and dot is None and self.gemm_eligible(kh, kw, ph, pw, inv.span, inv_main.up, flags)
            if wk32:
                flags |= ir.F_WK32
            if self.hilo:
                flags |= ir.F_HILO
            w_off = self.add_weights(("conv", wname, tuple(inv.segs), ep["out_name"], wk32, self.hilo),
                                     lambda: self.tile_weights(self.pack_conv_weights(w, ep["scale"], inv)[0], 32 if wk32 else ir.KT,
                                                               hilo=self.hilo))
            wgate = self.pending_wgate.get(op["in"]["Input"][0])
            if wgate is not None:
                # the input is an SE output whose gate multiply was left to its consumers (_gate_foldable): this image's
                # weights = W * gate (OP_WSCALE, fp16), read by conv_gemm_kernel through M tiles aligned to images (F_IMGW)
                assert (kh, kw, sh, sw, ph, pw) == (1, 1, 1, 1, 0, 0) and dot is None and not (flags & ~(ir.F_RES | ir.F_WK32))
                assert self.gemm_eligible(kh, kw, ph, pw, inv.span, inv_main.up, flags), outname
                wbuf = self.new_buf(inv.n, 1, 1, Kp * coutp, esize=2)
                wview = View(wbuf, 0, inv.n, 1, 1, [(0, Kp * coutp)], Kp * coutp)
                self.emit(ir.OP_WSCALE, ep["out_name"] + ":wgate", [wgate], wview, p={0: Kp, 1: coutp, 2: 32 if wk32 else ir.KT},
                          w_off=w_off)
                flags |= ir.F_IMGW
                while len(ins) < 2:
                    ins.append(None)
                ins.append(wview)
        b_off = self.add_weights(("convb", wname, ep["out_name"]), bias)
        if a.get("out_gate") is not None:
            # an SE block with shortcut folded into this 1x1 conv (_rewrite_se_laterals): out = conv * (1 + gate[n, c]) (+ residual)
            gv = self.resolve(a["out_gate"])
            # The kernel evaluates (acc + bias) * (1 + gate) + residual.  The rewritten conv's output IS the SE block's output, so every
            # affine / activation absorb_epilogue folded lies BEHIND the gate in the graph: (conv * (1 + g)) * s + b — which is not
            # (conv * s + b) * (1 + g).  Only the identity (and the residual add) may ride in this epilogue.
            affine_id = bool(np.all(ep["scale"] == 1.0) and not np.any(ep["shift"]) and ep["post_a"] == 1.0 and ep["post_b"] == 0.0)
            if (gv is None or (gv.h, gv.w) != (1, 1) or gv.c != cout or gv.segs != [(0, cout)] or gv.up or dot is not None or not affine_id
                    or flags & (ir.F_SRC2 | ir.F_IMGW | ir.F_PATCH | ir.F_COL | ir.F_STEM) or ep["act"] != ir.ACT_NONE or ep["act2"] != ir.ACT_NONE):
                raise GatedConvUnsupported(f"gated conv {outname}: the gate / layer form is not supported (gate {gv and (gv.h, gv.w, gv.c, gv.segs, gv.up)}, "
                                           f"flags {flags:#x}, act {ep['act']}/{ep['act2']}, identity affine behind the gate: {affine_id}, "
                                           f"dot {dot is not None})")
            flags |= ir.F_OGATE
            while len(ins) < 2:
                ins.append(None)
            ins.append(gv)
        if dot is not None:
            aux_off = self.add_weights(("dot1", dot["wname"], ep["out_name"]), dot["w"])
            self.emit(ir.OP_CONV, dot["out_name"], ins, dot["view"], flags=flags | ir.F_DOT1,
                      p={ir.P_KH: kh, ir.P_KW: kw, ir.P_SH: sh, ir.P_SW: sw, ir.P_PH: ph, ir.P_PW: pw,
                         ir.P_ACT: ep["act"], ir.P_ACT2: ep["act2"], ir.P_COUT: coutp, ir.P_KTOT: Kp,
                         ir.P_INSHIFT: inv_main.up, ir.P_RESSHIFT: resshift, ir.P_CINP: inv.span, ir.P_DOTACT: dot["act"],
                         ir.P_IN2SHIFT: in2shift},
                      f={ir.FS_ACT_A: ep["act_a"], ir.FS_ACT_B: ep["act_b"], ir.FS_POST_A: ep["post_a"],
                         ir.FS_POST_B: ep["post_b"], ir.FS_PRE_B: dot["b"]}, w_off=w_off, b_off=b_off,
                      aux_off=aux_off, out2=dot["view"])
            self.add_gmacs(inv.n * oh * ow * (cin * cout * kh * kw + cout) / 1e9)
            self.env[dot["out_name"]] = dot["view"]
            return
        self.emit(ir.OP_CONV, ep["out_name"], ins, out, flags=flags,
                  p={ir.P_KH: kh, ir.P_KW: kw, ir.P_SH: sh, ir.P_SW: sw, ir.P_PH: ph, ir.P_PW: pw,
                     ir.P_ACT: ep["act"], ir.P_ACT2: ep["act2"], ir.P_COUT: coutp, ir.P_KTOT: Kp,
                     ir.P_INSHIFT: inv_main.up, ir.P_RESSHIFT: resshift, ir.P_CINP: inv.span, ir.P_IN2SHIFT: in2shift,
                     ir.P_LO_OUT: out.buf.lo_off,
                     ir.P_LO_RES: (res.buf.lo_off if (res is not None and res.buf is not None and not res.up and res.coff == 0
                                                      and getattr(self, "chain", False)) else 0)},
                  f={ir.FS_ACT_A: ep["act_a"], ir.FS_ACT_B: ep["act_b"], ir.FS_POST_A: ep["post_a"],
                     ir.FS_POST_B: ep["post_b"]}, w_off=w_off, b_off=b_off)
        self.add_gmacs(inv.n * oh * ow * (cin // 2 if pair_in else cin) * cout * self.merged_gmac_credit.get(wname, kh * kw) / 1e9)
        self.env[ep["out_name"]] = out

    def lower_dwconv(self, i, inv, w, sh, sw, ph, pw):
        op = self.ops[i]
        outname = op["out"]["Output"][0]
        wname = op["in"]["Filter"][0]
        c, _, kh, kw = w.shape
        assert c == inv.c
        inv = self.need_dense(inv, f"depthwise conv {outname}")
        gate = self.pending_gate.pop(op["in"]["Input"][0], None)
        inv = self.materialize(inv, outname)
        ep = self.absorb_epilogue(outname, i, c, allow_res=False)
        oh = (inv.h + 2 * ph - kh) // sh + 1
        ow = (inv.w + 2 * pw - kw) // sw + 1
        cp = inv.span
        wk = np.zeros((kh * kw, cp), np.float32)
        wk[:, :c] = (w.astype(np.float64)[:, 0] * ep["scale"].reshape(-1, 1, 1)).reshape(c, kh * kw).T
        bias = np.zeros(cp, np.float32)
        bias[:c] = ep["shift"]
        out = self.alloc_out(ep["out_name"], inv.n, oh, ow, c, lo=self.wants_lo(ep["out_name"]))
        # the filter table the kernels read is fp32 [taps][cp]: fp16(w) — with hi + lo weights fp16(w) + fp16(w - fp16(w)), an exact
        # fp32 sum — so the depthwise kernels (VALU-bound) neither convert nor add weight halves per tap (round 4; the values are the
        # ones the fp16 tables of rounds 1-3 produced)
        wk_hi = wk.astype(np.float16)
        wk32 = wk_hi.astype(np.float32)
        if self.hilo:
            wk32 = wk32 + (wk.astype(np.float64) - wk_hi.astype(np.float64)).astype(np.float16).astype(np.float32)
        w_off = self.add_weights(("dw32", wname, ep["out_name"], self.hilo), wk32.reshape(-1))
        b_off = self.add_weights(("dwb", wname, ep["out_name"]), bias)
        self.emit(ir.OP_DWCONV, ep["out_name"], [inv] if gate is None else [inv, gate[0]], out,
                  flags=(0 if gate is None else (ir.F_GATE | gate[1])) | (ir.F_HILO if self.hilo else 0),
                  p={ir.P_KH: kh, ir.P_KW: kw, ir.P_SH: sh, ir.P_SW: sw, ir.P_PH: ph, ir.P_PW: pw,
                     ir.P_ACT: ep["act"], ir.P_LO_OUT: out.buf.lo_off,
                     # (P_LO_RES of a depthwise conv = the pair offset of its INPUT: both halves are filtered)
                     ir.P_LO_RES: inv.buf.lo_off if (gate is None and inv.coff == 0 and getattr(self, "chain", False)) else 0},
                  f={ir.FS_ACT_A: ep["act_a"], ir.FS_ACT_B: ep["act_b"], ir.FS_POST_A: ep["post_a"],
                     ir.FS_POST_B: ep["post_b"]}, w_off=w_off, b_off=b_off)
        self.add_gmacs(inv.n * oh * ow * c * kh * kw / 1e9)
        self.env[ep["out_name"]] = out

    def dwpw_eligible(self, i):
        """Structural half of try_lower_dwpw, without side effects: op i is a depthwise conv (3x3 / 5x5, stride 1 / 2, <= 96 channels)
        whose only reader — behind its own BN / activation — is a plain 1x1 conv with <= 192 couts."""
        if not (DWPW and self.hilo) or self.ragged or i in self.done or not self.live[i]:
            return False
        op = self.ops[i]
        if op["type"] not in ("conv2d", "depthwise_conv2d"):
            return False
        w = self.W[op["in"]["Filter"][0]]
        a = op["attrs"]
        groups = a.get("groups", 1)
        if not (op["type"] == "depthwise_conv2d" or (groups > 1 and groups == w.shape[0] and w.shape[1] == 1)):
            return False
        c, _, kh, kw = w.shape
        sh, sw = a["strides"]
        pads = a["paddings"]
        ph, pw = (pads[0], pads[1]) if len(pads) == 2 else (pads[0], pads[2])
        if kh != kw or kh not in DWPW_K or sh != sw or sh not in (1, 2) or ph != pw or ph != kh // 2 or c % 8 or c > 96:
            return False
        snapshot = set(self.done)
        try:
            ep_d = self.absorb_epilogue(op["out"]["Output"][0], i, c, allow_res=False)
        finally:
            self.done = snapshot
        dname = ep_d["out_name"]
        cons = self._live_consumers(dname)
        if len(cons) != 1 or dname in self.placement or dname in self.fetched_names or ep_d["act2"] != ir.ACT_NONE:
            return False
        o2 = self.ops[cons[0]]
        if o2["type"] != "conv2d" or o2["in"]["Input"][0] != dname or o2["attrs"].get("out_gate") is not None:
            return False
        w2 = self.W[o2["in"]["Filter"][0]]
        a2 = o2["attrs"]
        return (a2.get("groups", 1) == 1 and tuple(w2.shape[1:]) == (c, 1, 1) and list(a2["strides"]) == [1, 1] and not any(a2["paddings"])
                and rup(w2.shape[0], 8) <= 192)

    def try_lower_dwpw(self, i):
        """depthwise k x k conv whose only reader is a 1x1 stride-1 conv (the PP-LCNetV3 unit, the depthwise -> project half of a
        MobileNetV3 unit) in a hi + lo net -> ONE conv op (F_DWPRE, csrc/conv_dwpw.hip): the lane that needs 8 channels of a pixel as
        its MFMA B fragment COMPUTES them from the k x k neighbourhood (fp32, split into an fp16 hi + lo pair) instead of loading
        them.  The depthwise output — the widest tensor of the unit — is never written, read back or rounded."""
        if not self.dwpw_eligible(i):
            return False
        op = self.ops[i]
        w = self.W[op["in"]["Filter"][0]]
        a = op["attrs"]
        groups = a.get("groups", 1)
        if not (op["type"] == "depthwise_conv2d" or (op["type"] == "conv2d" and groups > 1 and groups == w.shape[0] and w.shape[1] == 1)):
            return False
        c, _, kh, kw = w.shape
        sh, sw = a["strides"]
        pads = a["paddings"]
        ph, pw = (pads[0], pads[1]) if len(pads) == 2 else (pads[0], pads[2])
        if kh != kw or kh not in (3, 5) or sh != sw or sh not in (1, 2) or ph != pw or c % 8 or c > 96:
            return False
        inname = op["in"]["Input"][0]
        inv = self.resolve(inname)
        if (inv is None or inv.tag != "nchw" or inv.parts is not None or inv.up or inv.segs != [(0, inv.c)] or inv.c != c or inv.buf.esize != 2
                or inv.coff % 8 or inname in self.pending_gate or inname in self.pending_wgate):
            return False
        snapshot = set(self.done)
        ep_d = self.absorb_epilogue(op["out"]["Output"][0], i, c, allow_res=False)
        dname = ep_d["out_name"]
        cons = self._live_consumers(dname)
        ok = len(cons) == 1 and dname not in self.placement and dname not in self.fetched_names and ep_d["act2"] == ir.ACT_NONE
        if ok:
            o2 = self.ops[cons[0]]
            ok = o2["type"] == "conv2d" and o2["in"]["Input"][0] == dname and o2["attrs"].get("out_gate") is None
        if ok:
            w2 = self.W[o2["in"]["Filter"][0]]
            a2 = o2["attrs"]
            ok = (a2.get("groups", 1) == 1 and tuple(w2.shape[1:]) == (c, 1, 1) and list(a2["strides"]) == [1, 1] and not any(a2["paddings"])
                  and rup(w2.shape[0], 8) <= 192)
        if not ok:
            self.done = snapshot
            return False
        j = cons[0]
        cout = int(w2.shape[0])
        oh = (inv.h + 2 * ph - kh) // sh + 1
        ow = (inv.w + 2 * pw - kw) // sw + 1
        ep = self.absorb_epilogue(o2["out"]["Output"][0], j, cout, out_dims=(inv.n, oh, ow))
        res = ep["res"]
        if res is not None and (res.up or (res.n, res.h, res.w, res.c) != (inv.n, oh, ow, cout) or res.segs != [(0, cout)]):
            self.done = snapshot
            return False
        self.done.add(i)
        self.done.add(j)
        coutp, ks = rup(cout, 8), rup(c, 16) // 16
        cp = ks * 16
        # the 1x1 weights: conv_pw's plain [Np][Kp] layout, hi table then lo table
        mat = np.zeros((coutp, cp), np.float64)
        mat[:cout, :c] = w2[:, :, 0, 0].astype(np.float64) * ep["scale"].reshape(-1, 1)
        bias = np.zeros(coutp, np.float32)
        bias[:cout] = ep["shift"]
        # the depthwise table: [k*k + 1][Kp] fp32 (BN / affine folded; last row = bias), behind 8 header words
        k2 = kh * kw
        tab = np.zeros((k2 + 1, cp), np.float32)
        tab[:k2, :c] = (w.astype(np.float64)[:, 0] * ep_d["scale"].reshape(-1, 1, 1)).reshape(c, k2).T
        tab[k2, :c] = ep_d["shift"]
        hdr = np.zeros(8, np.float32)
        hdr.view(np.int32)[:4] = [kh, sh, ph, ep_d["act"]]
        hdr[4:] = [ep_d["act_a"], ep_d["act_b"], ep_d["post_a"], ep_d["post_b"]]
        wname, dwname = o2["in"]["Filter"][0], op["in"]["Filter"][0]
        w_off = self.add_weights(("dwpw_w", wname, ep["out_name"]), lambda: self.pw_weights(mat, True))
        b_off = self.add_weights(("dwpw_b", wname, ep["out_name"]), bias)
        aux_off = self.add_weights(("dwpw_t", dwname, dname), np.concatenate([hdr, tab.reshape(-1)]))
        out = self.alloc_out(ep["out_name"], inv.n, oh, ow, cout, lo=self.wants_lo(ep["out_name"]))
        flags = ir.F_PW | ir.F_HILO | ir.F_DWPRE
        ins = [inv]
        if res is not None:
            flags |= ir.F_RES
            ins.append(res)
        self.emit(ir.OP_CONV, ep["out_name"], ins, out, flags=flags,
                  p={ir.P_KH: kh, ir.P_KW: kw, ir.P_SH: sh, ir.P_SW: sw, ir.P_PH: ph, ir.P_PW: pw, ir.P_ACT: ep["act"], ir.P_ACT2: ep["act2"],
                     ir.P_COUT: coutp, ir.P_KTOT: cp, ir.P_INSHIFT: 0, ir.P_RESSHIFT: 0, ir.P_CINP: inv.span,
                     ir.P_LO_OUT: out.buf.lo_off,
                     ir.P_LO_IN: inv.buf.lo_off if (inv.coff == 0 and getattr(self, "chain_lo", True)) else 0,
                     ir.P_LO_RES: (res.buf.lo_off if (res is not None and res.buf is not None and res.coff == 0) else 0)},
                  f={ir.FS_ACT_A: ep["act_a"], ir.FS_ACT_B: ep["act_b"], ir.FS_POST_A: ep["post_a"], ir.FS_POST_B: ep["post_b"]},
                  w_off=w_off, b_off=b_off, aux_off=aux_off)
        self.add_gmacs(inv.n * oh * ow * (c * k2 + c * cout) / 1e9)
        self.env[ep["out_name"]] = out
        return True

    def lower_linear(self, i):
        """matmul_v2 with a parameter RHS = 1x1 conv over [B,1,T,C]."""
        op = self.ops[i]
        x = self.resolve(op["in"]["X"][0])
        wname = op["in"]["Y"][0]
        w = self.W[wname]      # [in,out]
        assert x is not None and x.tag in ("btc", "tbc"), (i, x and x.tag)
        assert not op["attrs"].get("trans_x", False) and not op["attrs"].get("trans_y", False)
        outname = op["out"]["Out"][0]
        cin, cout = w.shape
        assert cin == x.c
        # attention qkv projection?  (linear -> reshape [0,-1,3,heads,hd])
        ep = self.absorb_epilogue(outname, i, cout, out_dims=(x.n, x.h, x.w))
        w4 = w.T.reshape(cout, cin, 1, 1)
        coutp, Kp = rup(cout, 8), rup(x.span, ir.KT)
        bias = np.zeros(coutp, np.float32)
        bias[:cout] = ep["shift"]
        # logits that feed the final class softmax stay fp32 (fp16 ulp at |logit|~10 is 8e-3: too coarse for the
        # 1e-3 tolerance on the recogniser's probabilities)
        cons = self._live_consumers(ep["out_name"])
        to_softmax = len(cons) == 1 and self.ops[cons[0]]["type"] == "softmax"
        flags = 0
        if to_softmax:
            ob = self.new_buf(x.n, x.h, x.w, coutp, esize=4)
            out = View(ob, 0, x.n, x.h, x.w, [(0, cout)], coutp)
            flags |= ir.F_OUT_F32
        else:
            out = self.alloc_out(ep["out_name"], x.n, x.h, x.w, cout)
        out.tag = x.tag
        ins = [x]
        if ep["res"] is not None:
            r = ep["res"]
            assert (r.n, r.h, r.w, r.c) == (x.n, x.h, x.w, cout) and r.up == 0
            ins.append(r)
            flags |= ir.F_RES
        w_off = self.add_weights(("lin", wname, tuple(x.segs), ep["out_name"]),
                                 lambda: self.tile_weights(self.pack_conv_weights(w4, ep["scale"], x)[0]))
        b_off = self.add_weights(("linb", wname, ep["out_name"]), bias)
        self.emit(ir.OP_CONV, ep["out_name"], ins, out, flags=flags,
                  p={ir.P_KH: 1, ir.P_KW: 1, ir.P_SH: 1, ir.P_SW: 1, ir.P_PH: 0, ir.P_PW: 0,
                     ir.P_ACT: ep["act"], ir.P_ACT2: ep["act2"], ir.P_COUT: coutp, ir.P_KTOT: Kp,
                     ir.P_INSHIFT: 0, ir.P_RESSHIFT: 0, ir.P_CINP: x.span},
                  f={ir.FS_ACT_A: ep["act_a"], ir.FS_ACT_B: ep["act_b"], ir.FS_POST_A: ep["post_a"],
                     ir.FS_POST_B: ep["post_b"]}, w_off=w_off, b_off=b_off)
        self.add_gmacs(x.n * x.h * x.w * cin * cout / 1e9)
        self.env[ep["out_name"]] = out

    # -------------------------------------------------------------------------------------------- attention
    def try_lower_attention(self, i):
        """reshape2(qkv)[B,T,3,h,d] -> transpose -> slices -> scale -> q.kT -> softmax -> .v -> transpose ->
        reshape [B,T,C]  ==> one OP_ATTN.  Returns True when matched."""
        op = self.ops[i]
        src = self.resolve(op["in"]["X"][0])
        if src is None or src.tag != "btc":
            return False
        cons = self._live_consumers(op["out"]["Out"][0])
        if len(cons) != 1 or self.ops[cons[0]]["type"] != "transpose2" or \
                list(self.ops[cons[0]]["attrs"]["axis"]) != [2, 0, 3, 1, 4]:
            return False
        C3 = src.c
        assert C3 % 3 == 0
        C = C3 // 3
        # walk forward collecting the block until the reshape back to [B,T,C]
        j = cons[0]
        heads = None
        scale = None
        visited = [i, j]
        frontier = [self.ops[j]["out"]["Out"][0]]
        end_name = None
        seen = set()
        while frontier:
            nm = frontier.pop()
            for k in self._live_consumers(nm):
                if k in seen:
                    continue
                seen.add(k)
                o = self.ops[k]
                visited.append(k)
                t = o["type"]
                if t == "scale":
                    scale = o["attrs"]["scale"]
                if t == "reshape2" and k != i:
                    end_name = o["out"]["Out"][0]
                    continue
                assert t in ("slice", "scale", "transpose2", "matmul_v2", "softmax", "dropout", "shape",
                             "fill_constant"), (t, k)
                frontier.append(o["out"]["Out"][0])
        assert end_name is not None and scale is not None
        hd = int(round(1.0 / (scale * scale)))
        heads = C // hd
        assert heads * hd == C, (C, hd)
        for k in visited:
            self.done.add(k)
        # shape-tensor helper ops feeding the reshapes
        qkv = self.materialize(src)
        out = self.alloc_out(end_name, qkv.n, qkv.h, qkv.w, C)
        out.tag = "btc"
        self.emit(ir.OP_ATTN, end_name, [qkv], out, p={ir.P_HEADS: heads, ir.P_HDIM: hd}, f={ir.FS_SCALE: scale})
        self.env[end_name] = out
        return True

    # -------------------------------------------------------------------------------------------- other ops
    def lower_pool(self, i):
        op = self.ops[i]
        a = op["attrs"]
        x = self.resolve(op["in"]["X"][0])
        assert x is not None and x.tag == "nchw"
        name = op["out"]["Out"][0]
        x = self.materialize(x, name)
        if a.get("adaptive", False) or a.get("global_pooling", False):
            assert a["pooling_type"] == "avg" and (a.get("global_pooling", False) or list(a["ksize"]) == [1, 1])
            out = self.alloc_out(name, x.n, 1, 1, x.c)
            out.segs, out.span = list(x.segs), x.span
            if out.buf.ld != x.span:
                out = View(self.new_buf(x.n, 1, 1, x.span), 0, x.n, 1, 1, list(x.segs), x.span)
            # enough blocks to hide the load latency on small maps too (one block streams 64 channels of h*w/splits pixels)
            splits = max(1, min(64, (x.h * x.w) // 256))
            if self.ragged:
                splits = x.h          # gap_rows_kernel: one partial sum per row, a summation order that ignores the tensor's width
            sb = self.new_buf(x.n, splits, 1, x.span, esize=4)
            scratch = View(sb, 0, x.n, splits, 1, [(0, x.span)], x.span)
            self.emit(ir.OP_GAP, name, [x, None, scratch], out)
            self.env[name] = out
            return
        kh, kw = a["ksize"]
        sh, sw = a["strides"]
        pads = a["paddings"]
        ph, pw = (pads[0], pads[1]) if len(pads) == 2 else (pads[0], pads[2])
        ceil = bool(a.get("ceil_mode", False))

        def osz(n, k, s, p):
            if ceil:
                o = -(-(n + 2 * p - k) // s) + 1
                if (o - 1) * s >= n + p:
                    o -= 1
                return o
            return (n + 2 * p - k) // s + 1
        oh, ow = osz(x.h, kh, sh, ph), osz(x.w, kw, sw, pw)
        x = self.need_dense(x, f"pool {name}")
        out = self.alloc_out(name, x.n, oh, ow, x.c)
        self.emit(ir.OP_POOL, name, [x], out,
                  p={ir.P_KH: kh, ir.P_KW: kw, ir.P_SH: sh, ir.P_SW: sw, ir.P_PH: ph, ir.P_PW: pw,
                     ir.P_POOL_MAX: int(a["pooling_type"] == "max"), ir.P_POOL_CEIL: int(ceil),
                     ir.P_POOL_EXCL: int(a.get("exclusive", True))})
        self.env[name] = out

    def lower_binary(self, i):
        op = self.ops[i]
        t = op["type"]
        xn, yn = op["in"]["X"][0], op["in"]["Y"][0]
        name = op["out"]["Out"][0]
        if self.is_param(xn) or self.is_param(yn):
            pn, tn = (xn, yn) if self.is_param(xn) else (yn, xn)
            s = self._scalar_param(pn)
            x = self.resolve(tn)
            assert s is not None and x is not None, f"unfused non-scalar param {t} at op {i}"
            x = self.materialize(x, name)
            out = self.alloc_out(name, x.n, x.h, x.w, x.c)
            out.tag = x.tag
            a, b = (s, 0.0) if t == "elementwise_mul" else (1.0, s)
            self.emit(ir.OP_UNARY, name, [x], out, p={0: ir.ACT_NONE},
                      f={ir.FS_PRE_A: a, ir.FS_PRE_B: b, ir.FS_POST_A: 1.0, ir.FS_POST_B: 0.0})
            self.env[name] = out
            return
        x, y = self.resolve(xn), self.resolve(yn)
        assert x is not None and y is not None, (i, xn, yn)
        if t == "elementwise_mul":
            # SE gate: big [N,H,W,C] * gate [N,1,1,C]; optionally followed by "+ big" (residual SE)
            if (y.h, y.w) == (1, 1):
                big, gate, bigname = x, y, xn
            else:
                big, gate, bigname = y, x, yn
            assert (gate.h, gate.w) == (1, 1) and gate.c == big.c, (i, x, y)
            big = self.materialize(big, name)
            flags = 0
            outname = name
            cons = self._live_consumers(name)
            if len(cons) == 1 and name not in self.placement:
                o2 = self.ops[cons[0]]
                if o2["type"] == "elementwise_add" and bigname in (o2["in"]["X"][0], o2["in"]["Y"][0]):
                    flags |= ir.F_RES
                    outname = o2["out"]["Out"][0]
                    self.done.add(cons[0])
            big, gate = self.need_dense(big, f"gate multiply {name}"), self.need_dense(gate, f"gate multiply {name}")
            # SE gate whose only consumer is a depthwise conv (the stage transitions of the HGNet recognisers): the conv
            # applies the gate on load and the scaled tensor is never written
            cons2 = self._live_consumers(outname)
            if GATE_DW and len(cons2) == 1 and outname not in self.placement and outname not in self.fetched_names:
                o2 = self.ops[cons2[0]]
                if o2["type"] in ("conv2d", "depthwise_conv2d") and o2["in"]["Input"][0] == outname:
                    w2 = self.W[o2["in"]["Filter"][0]]
                    g2 = o2["attrs"].get("groups", 1)
                    if (o2["type"] == "depthwise_conv2d" or (g2 > 1 and g2 == w2.shape[0])) and w2.shape[1] == 1 and w2.shape[0] == big.c:
                        self.pending_gate[outname] = (gate, flags)
                        self.env[outname] = big
                        return
            if GATE_FOLD and self._gate_foldable(outname, big, flags):
                # every consumer applies the gate itself: the depthwise conv on load (F_GATE), the 1x1 convs through per-image
                # weights W * gate (OP_WSCALE + F_IMGW) — the scaled tensor (a full read + write of the stage output) is never made
                for j in cons2:
                    if self._is_dw(self.ops[j], big.c):
                        self.pending_gate[outname] = (gate, flags)
                    else:
                        self.pending_wgate[outname] = gate
                self.env[outname] = big
                return
            if GATE_CONCAT and self._only_feeds_concat(outname) and big.c % 8 == 0 and gate.c == big.c:
                # the SE output is read by a concat only (directly or through a nearest up-sampling): the copy into the concat slot
                # multiplies (lower_concat, gated OP_RESIZE) — no separate scale pass, one rounding instead of two
                self.env[outname] = View(big.buf, big.coff, big.n, big.h, big.w, list(big.segs), big.span, big.up, big.tag, gate=(gate, flags))
                return
            out = self.alloc_out(outname, big.n, big.h, big.w, big.c)
            self.emit(ir.OP_SCALE, outname, [big, gate], out, flags=flags)
            self.env[outname] = out
            return
        # add of two activation tensors (y may be a virtual upsample of a coarser map)
        if x.up and not y.up:
            x, y = y, x
        x = self.materialize(x, name)
        assert (x.n, x.h, x.w, x.c) == (y.n, y.h, y.w, y.c), (i, x, y)
        x, y = self.need_dense(x, f"add {name}"), self.need_dense(y, f"add {name}")
        act = ir.ACT_NONE
        outname = name
        cons = self._live_consumers(name)
        if len(cons) == 1 and name not in self.placement and self.ops[cons[0]]["type"] == "relu":
            act = ir.ACT_RELU
            outname = self.ops[cons[0]]["out"]["Out"][0]
            self.done.add(cons[0])
        out = self.alloc_out(outname, x.n, x.h, x.w, x.c)
        out.tag = x.tag
        self.emit(ir.OP_BINARY, outname, [x, y], out, p={ir.P_BIN_MUL: 0, ir.P_BIN_SHIFT: y.up, ir.P_BIN_ACT: act})
        self.env[outname] = out

    def _only_feeds_concat(self, name):
        """`name` is read by exactly one op: a non-virtual concat, or a nearest_interp whose only reader is one."""
        cons = self._live_consumers(name)
        if len(cons) != 1 or name in self.fetched_names:
            return False
        o = self.ops[cons[0]]
        if o["type"] == "nearest_interp_v2":
            nm = o["out"]["Out"][0]
            cons = self._live_consumers(nm)
            if len(cons) != 1 or nm in self.fetched_names:
                return False
            o = self.ops[cons[0]]
        return o["type"] == "concat" and o["out"]["Out"][0] not in self.virtual_concats and o["attrs"].get("axis") == 1

    def lower_unary(self, i):
        op = self.ops[i]
        t = op["type"]
        a = op["attrs"]
        x = self.resolve(op["in"]["X"][0])
        name = op["out"]["Out"][0]
        x = self.need_dense(self.materialize(x, name), f"{t} {name}")
        out = self.alloc_out(name, x.n, x.h, x.w, x.c)
        out.tag = x.tag
        f = {ir.FS_PRE_A: 1.0, ir.FS_PRE_B: 0.0, ir.FS_POST_A: 1.0, ir.FS_POST_B: 0.0}
        act = ir.ACT_NONE
        if t == "scale":
            if a.get("bias_after_scale", True):
                f[ir.FS_PRE_A], f[ir.FS_PRE_B] = a["scale"], a.get("bias", 0.0)
            else:
                f[ir.FS_PRE_A], f[ir.FS_PRE_B] = a["scale"], a.get("bias", 0.0) * a["scale"]
        else:
            act = _ACTS[t]
            if t == "hard_sigmoid":
                f[ir.FS_ACT_A], f[ir.FS_ACT_B] = a["slope"], a["offset"]
        self.emit(ir.OP_UNARY, name, [x], out, p={0: act}, f=f)
        self.env[name] = out

    def lower_concat(self, i):
        op = self.ops[i]
        name = op["out"]["Out"][0]
        assert op["attrs"]["axis"] == 1
        ins = [self.resolve(n) for n in op["in"]["X"]]
        assert all(v is not None for v in ins), (i, op["in"]["X"])
        v0 = ins[0]
        if name in self.virtual_concats and all(v.segs == [(0, v.c)] for v in ins):
            segs, off = [], 0
            for v in ins:
                segs.append((off, v.c))
                off += v.span
            self.env[name] = View(None, 0, v0.n, v0.h, v0.w, merge_segs(segs), off, 0, "nchw", parts=list(ins))
            return
        lay = self._concat_buf(name, v0.n, v0.h, v0.w)
        b = lay["buf"]
        segs = []
        copies = []                     # (input view, destination channel offset, channels) of the inputs that are not in place
        for nm, v, (off, c) in zip(op["in"]["X"], ins, lay["offs"]):
            assert (v.n, v.h, v.w, v.c) == (b.n, b.h, b.w, c), (name, nm, v, c)
            if not (v.buf is b and v.coff == off and v.up == 0 and v.gate is None):
                if v.parts is not None:
                    v = self.materialize(v, name + ":" + nm)
                if v.gate is not None or (v.segs == [(0, v.c)] and v.c % 8 == 0):
                    copies.append((nm, v, off, c))
                else:
                    # a multi-segment input (a nested concat whose parts are not multiples of 8 channels) is copied piece by piece;
                    # every piece must land on an 8-channel boundary of the destination (the copy kernel moves 8-channel groups)
                    done_c = 0
                    for st, cnt in v.segs:
                        if done_c % 8:
                            raise UnsupportedGraph(f"concat {name}: input {nm} has channel segments {v.segs} that do not fall on "
                                                   "8-channel boundaries")
                        piece = View(v.buf, v.coff + st, v.n, v.h, v.w, [(0, cnt)], rup(cnt, 8), v.up, v.tag)
                        dst = View(b, off + done_c, v.n, v.h, v.w, [(0, cnt)], rup(cnt, 8))
                        self.emit(ir.OP_RESIZE, name + ":" + nm, [piece], dst, p={0: v.up})
                        done_c += cnt
            segs.append((off, c))
        # dense inputs: plain / gated copies; two inputs with ADJACENT slots share a launch (2 x C contiguous channels per pixel)
        k = 0
        while k < len(copies):
            nm, v, off, c = copies[k]
            pair = (k + 1 < len(copies) and copies[k + 1][2] == off + c and c % 8 == 0
                    and (v.gate is None) == (copies[k + 1][1].gate is None)
                    and (v.gate is None or v.gate[1] == copies[k + 1][1].gate[1]))
            src = View(v.buf, v.coff, v.n, v.h, v.w, [(0, c)], rup(c, 8), v.up, v.tag)
            fl = 0 if v.gate is None else (ir.F_GATE | (v.gate[1] & ir.F_RES))
            if pair:
                nm2, v2, off2, c2 = copies[k + 1]
                src2 = View(v2.buf, v2.coff, v2.n, v2.h, v2.w, [(0, c2)], rup(c2, 8), v2.up, v2.tag)
                dst = View(b, off, v.n, v.h, v.w, [(0, c + c2)], rup(c + c2, 8))
                self.emit(ir.OP_RESIZE, name + ":" + nm + "+" + nm2, [src, v.gate[0] if v.gate else None, src2], dst, flags=fl | ir.F_SRC2,
                          p={0: v.up, 1: v2.up}, out2=v2.gate[0] if v2.gate else None)
                k += 2
            else:
                dst = View(b, off, v.n, v.h, v.w, [(0, c)], rup(c, 8))
                if fl:
                    self.emit(ir.OP_RESIZE, name + ":" + nm, [src, v.gate[0]], dst, flags=fl, p={0: v.up})
                else:
                    self.emit(ir.OP_RESIZE, name + ":" + nm, [src], dst, p={0: v.up})
                k += 1
        out = View(b, 0, v0.n, v0.h, v0.w, merge_segs(segs), b.ld)
        self.env[name] = out

    def lower_layernorm(self, i):
        op = self.ops[i]
        x = self.resolve(op["in"]["X"][0])
        assert x.tag == "btc" and x.up == 0 and x.c == x.span
        name = op["out"]["Y"][0]
        g = self.W[op["in"]["Scale"][0]].astype(np.float32).reshape(-1)
        b = self.W[op["in"]["Bias"][0]].astype(np.float32).reshape(-1)
        out = self.alloc_out(name, x.n, x.h, x.w, x.c)
        out.tag = "btc"
        w_off = self.add_weights(("ln", op["in"]["Scale"][0]), np.concatenate([g, b]))
        self.emit(ir.OP_LAYERNORM, name, [x], out, f={ir.FS_EPS: op["attrs"]["epsilon"]}, w_off=w_off)
        self.env[name] = out

    def lower_softmax_out(self, i):
        """Final class softmax -> (probs f32 [B,T,C] optional, argmax i32 [B,T], maxp f32 [B,T])."""
        op = self.ops[i]
        x = self.resolve(op["in"]["X"][0])
        assert x.tag == "btc" and x.up == 0
        name = op["out"]["Out"][0]
        ncls = x.c
        B, T = x.n, x.w
        pb = self.new_buf(B, 1, T, ncls, esize=4, ext=len(self.outputs) + 1) if self.want_probs else None
        if pb is not None:
            self.outputs.append(dict(name=name, kind="probs", n=B, h=1, w=T, c=ncls, ld=ncls, esize=4))
        ib = self.new_buf(B, 1, T, 2, esize=4, ext=len(self.outputs) + 1)
        self.outputs.append(dict(name=name + ":idx_maxp", kind="idx_maxp", n=B, h=1, w=T, c=2, ld=2, esize=4))
        pv = View(pb, 0, B, 1, T, [(0, ncls)], ncls) if pb is not None else None
        iv = View(ib, 0, B, 1, T, [(0, 2)], 2)
        self.emit(ir.OP_SOFTMAX, name, [x], iv, p={ir.P_NCLS: ncls}, out2=pv)
        self._out_level = x.buf.wl
        self.env[name] = iv
        self._fetched = name

    @staticmethod
    def lstm_gate_order(H, waves):
        """Channel order of the gate pre-activations handed to the MFMA LSTM kernels: new[w * 4U + g * U + u] = old[g * H + w * U + u],
        U = H / waves hidden units per wave."""
        U = H // waves
        return np.arange(4 * H).reshape(4, waves, U).transpose(1, 0, 2).reshape(-1)

    @staticmethod
    def lstm_fragments16(w_hh):
        """The 16-wave form of lstm_fragments (lstm_mfma16_kernel): wave w owns units 16w .. 16w+15; tile 0 = [gate i | gate f],
        tile 1 = [gate g | gate o] (rows 0-15 | 16-31); stream order [wave 16][k-slice 16][tile 2][k-half 2][row 32][8]."""
        H = w_hh.shape[1]
        assert w_hh.shape == (4 * H, H) and H == 256
        w6 = w_hh.reshape(2, 2, 16, 16, 16, 2, 8)                 # [tile][gate-in-tile][wave][unit][slice][k-half][j]
        return np.ascontiguousarray(w6.transpose(2, 4, 0, 5, 1, 3, 6)).astype(np.float16).reshape(-1)    # [wave][slice][tile][k-half][(gate, unit) = row][j]

    @staticmethod
    def lstm_fragments(w_hh):
        """W_hh [4H, H] (gate order i, f, g, o; H = 256) -> fp16 in the order lstm_mfma_kernel streams it: wave w owns hidden
        units 32w .. 32w+31; per (wave, 16-deep k slice, gate) — the order of the stream — one MFMA A fragment = [lane 64][8]: row = lane & 31 (unit
        32w + row of that gate), k = 16 s + 8 (lane >> 5) + j."""
        H = w_hh.shape[1]
        assert w_hh.shape == (4 * H, H) and H == 256
        w5 = w_hh.reshape(4, 8, 32, 16, 2, 8)                     # [gate][wave][row][slice][k-half][j]
        return np.ascontiguousarray(w5.transpose(1, 3, 0, 4, 2, 5)).astype(np.float16).reshape(-1)       # [wave][slice][gate][k-half][row][j]

    def lower_rnn(self, i):
        op = self.ops[i]
        a = op["attrs"]
        assert a["mode"] == "LSTM"
        x = self.resolve(op["in"]["Input"][0])
        assert x.tag == "tbc", x.tag
        x = self.materialize(x)
        H = a["hidden_size"]
        ndir = 2 if a["is_bidirec"] else 1
        L = a["num_layers"]
        wl = op["in"]["WeightList"]
        ncell = L * ndir
        cur = x
        name = op["out"]["Out"][0]
        mfma = LSTM_MFMA and H == 256
        for layer in range(L):
            outb = self.new_buf(cur.n, 1, cur.w, ndir * H)
            gate_views, frag = [], []
            for d in range(ndir):
                c = layer * ndir + d
                w_ih = self.W[wl[2 * c]].astype(np.float64)       # [4H,in]
                w_hh = self.W[wl[2 * c + 1]].astype(np.float32)   # [4H,H]
                b = (self.W[wl[2 * ncell + 2 * c]].astype(np.float64) +
                     self.W[wl[2 * ncell + 2 * c + 1]].astype(np.float64))
                if mfma:
                    # the gate pre-activations leave the projection GEMM in the order the recurrent kernel reads them: the four
                    # gates of a wave's hidden units side by side ([wave][gate][unit in wave]: two whole cache lines per (wave,
                    # sample) instead of four half lines a kilobyte apart) — a permutation of the GEMM's output channels
                    gperm = self.lstm_gate_order(H, LSTM_WAVES)
                    w_ih, b = w_ih[gperm], b[gperm]
                # input projection for all T as one GEMM (fp32 output to keep gate pre-activations exact-ish)
                mat, coutp, Kp = self.pack_conv_weights(w_ih.reshape(4 * H, -1, 1, 1), np.ones(4 * H), cur)
                gb = self.new_buf(cur.n, 1, cur.w, 4 * H, esize=4)
                gates = View(gb, 0, cur.n, 1, cur.w, [(0, 4 * H)], 4 * H)
                key = f"{name}:l{layer}d{d}"
                w_off = self.add_weights(("lstm_ih", wl[2 * c], LSTM_WAVES if mfma else 0), self.tile_weights(mat))
                b_off = self.add_weights(("lstm_b", wl[2 * c], LSTM_WAVES if mfma else 0), b.astype(np.float32))
                self.emit(ir.OP_CONV, key + ":proj", [cur], gates, flags=ir.F_OUT_F32,
                          p={ir.P_KH: 1, ir.P_KW: 1, ir.P_SH: 1, ir.P_SW: 1, ir.P_PH: 0, ir.P_PW: 0,
                             ir.P_ACT: 0, ir.P_ACT2: 0, ir.P_COUT: coutp, ir.P_KTOT: Kp, ir.P_INSHIFT: 0,
                             ir.P_RESSHIFT: 0, ir.P_CINP: cur.span},
                          f={ir.FS_POST_A: 1.0}, w_off=w_off, b_off=b_off)
                self.add_gmacs(cur.n * cur.w * cur.c * 4 * H / 1e9)
                if mfma:
                    gate_views.append(gates)
                    frag.append((wl[2 * c + 1], w_hh))
                    continue
                whh_off = self.add_weights(("lstm_hh", wl[2 * c + 1]), w_hh.T.copy().astype(np.float16))  # [H][4H]
                ov = View(outb, d * H, cur.n, 1, cur.w, [(0, H)], H)
                self.emit(ir.OP_LSTM, key, [gates], ov, p={ir.P_HID: H, ir.P_REVERSE: d}, w_off=whh_off)
                self.add_gmacs(cur.n * cur.w * H * 4 * H / 1e9)
            if mfma:
                # the recurrence of every direction of the layer in ONE launch (csrc/lstm.hip): batch-shared MFMA GEMM per step
                whh_off = self.add_weights(("lstm_hh_mfma", LSTM_WAVES) + tuple(nm for nm, _ in frag),
                                           lambda: np.concatenate([(self.lstm_fragments16 if LSTM_WAVES == 16 else self.lstm_fragments)(w)
                                                                   for _, w in frag]))
                ov = View(outb, 0, cur.n, 1, cur.w, [(0, ndir * H)], ndir * H)
                self.emit(ir.OP_LSTM, f"{name}:l{layer}", gate_views, ov, flags=ir.F_LSTM_MFMA,
                          p={ir.P_HID: H, ir.P_REVERSE: 2 if ndir == 2 else 0, 2: LSTM_WAVES}, w_off=whh_off)
                self.add_gmacs(ndir * cur.n * cur.w * H * 4 * H / 1e9)
            cur = View(outb, 0, cur.n, 1, cur.w, [(0, ndir * H)], ndir * H, 0, "tbc")
        self.env[name] = cur

    # -------------------------------------------------------------------------------------------- driver
    def compile(self) -> Program:
        N, H, Wd = self.N, self.H, self.Wd
        inb = self.new_buf(N, H, Wd, 8, ext=0)
        if self.ragged:
            inb.wl = 0
        feed_c = self.fold_input_norm() if self.input_norm is not None else 3
        if self.fuse_preprocess and self.input_norm is None:
            raise UnsupportedGraph("fuse_preprocess needs input_norm (the stem must carry the normalisation)")
        self._u8_fused = False
        for i, op in enumerate(self.ops):
            if i in self.done or not self.live[i]:
                continue
            t = op["type"]
            if t == "feed":
                self.env[op["out"]["Out"][0]] = View(inb, 0, N, H, Wd, [(0, feed_c)], 8)
            elif t == "fetch":
                self._lower_fetch(i)
            elif t in ("conv2d", "depthwise_conv2d", "conv2d_transpose"):
                if not self.try_lower_dwpw(i) and not self.try_lower_chain(i) and not self.try_lower_head_tail(i):
                    self.lower_conv(i)
            elif t == "batch_norm":
                raise NotImplementedError(f"stand-alone batch_norm at op {i}")
            elif t == "pool2d":
                self.lower_pool(i)
            elif t in ("elementwise_add", "elementwise_mul"):
                self.lower_binary(i)
            elif t == "concat":
                self.lower_concat(i)
            elif t == "matmul_v2" or t == "matmul":
                assert self.is_param(op["in"]["Y"][0]), f"free matmul outside attention at op {i}"
                self.lower_linear(i)
            elif t == "layer_norm":
                self.lower_layernorm(i)
            elif t == "softmax":
                self.lower_softmax_out(i)
            elif t == "rnn":
                self.lower_rnn(i)
            elif t == "reshape2":
                if not self.try_lower_attention(i):
                    assert self._lower_virtual(i), i
            elif t in _ACTS or (t == "scale" and not self._scale_is_noop(op)):
                self.lower_unary(i)
            elif t in _VIRTUAL or t == "scale" or t == "slice":
                if t == "slice":
                    self.done.add(i)      # only shape-tensor plumbing reaches here
                    continue
                assert self._lower_virtual(i), (i, t)
            else:
                raise NotImplementedError(f"{t} at op {i}")
        if self.fuse_preprocess:
            readers = [o for o in self.ir_ops if any(v is not None and v.buf is not None and v.buf.ext == 0 for v in o["ins"])]
            if not self._u8_fused or len(readers) != 1 or not (readers[0]["flags"] & ir.F_U8SRC):
                raise UnsupportedGraph("fuse_preprocess: the plan input is not read by exactly one 3x3 stem conv (conv_stem_kernel)")
        return self._finish()

    def _lower_fetch(self, i):
        op = self.ops[i]
        name = op["in"]["X"][0]
        v = self.resolve(name)
        if v.buf.ext is not None:
            return      # softmax head already wrote external outputs
        # dense fp16 map [N,H,W,c]: copy/convert into an external fp32 buffer, channel-exact
        v = self.need_dense(self.materialize(v, name), f"fetch of {name}")
        ob = self.new_buf(v.n, v.h, v.w, v.c, esize=4, ext=len(self.outputs) + 1)
        self.outputs.append(dict(name=name, kind="map", n=v.n, h=v.h, w=v.w, c=v.c, ld=v.c, esize=4))
        ov = View(ob, 0, v.n, v.h, v.w, [(0, v.c)], v.c)
        self.emit(ir.OP_UNARY, "fetch:" + name, [v], ov, flags=ir.F_OUT_F32, p={0: ir.ACT_NONE},
                  f={ir.FS_PRE_A: 1.0, ir.FS_PRE_B: 0.0, ir.FS_POST_A: 1.0, ir.FS_POST_B: 0.0})

    def _allocate(self):
        """First-fit offsets by liveness for workspace buffers."""
        ws = [b for b in self.bufs if b.ext is None and b.last >= 0]
        ws.sort(key=lambda b: (b.first, -b.nbytes))
        placed = []   # (offset, size, first, last)
        total = 0
        for b in ws:
            size = rup(b.nbytes, 256)
            cands = sorted((p for p in placed if not self.reuse or not (p[3] < b.first or p[2] > b.last)),
                           key=lambda p: p[0])
            off = 0
            for p in cands:
                if off + size <= p[0]:
                    break
                off = max(off, p[0] + p[1])
            b.offset = off
            placed.append((off, size, b.first, b.last))
            total = max(total, off + size)
        return total

    def _finish(self) -> Program:
        ws_bytes = self._allocate()
        recs = np.zeros(len(self.ir_ops), dtype=ir.OP_DT)
        names = []
        for k, o in enumerate(self.ir_ops):
            r = recs[k]
            r["kind"] = o["kind"]
            r["flags"] = o["flags"]
            for idx, val in o["p"].items():
                r["p"][idx] = val
            for idx, val in o["f"].items():
                r["f"][idx] = val
            slots = ["in0", "in1", "in2"]
            for s, v in zip(slots, o["ins"]):
                if v is not None:
                    r[s] = self._final_view(v)
            r["out"] = self._final_view(o["out"])
            if o["out2"] is not None:
                r["out2"] = self._final_view(o["out2"])
            r["w_off"], r["b_off"], r["aux_off"] = o["w_off"], o["b_off"], o["aux_off"]
            names.append(o["name"])
        return Program(ops=recs, weights=self.store, ws_bytes=int(ws_bytes), in_shape=(self.N, self.H, self.Wd, 8),
                       outputs=self.outputs, names=names, gmacs=self.gmacs,
                       op_gmacs=[o["gmac"] for o in self.ir_ops],
                       wlevels=list(self.wlevels) if self.ragged else None, out_level=getattr(self, "_out_level", 0) or 0)

    def _final_view(self, v: View):
        r = self.vrec(v)
        b = v.buf
        if b.ext is None:
            r["arena"] = ir.ARENA_WS
            r["off"] = b.offset + v.coff * b.esize
        else:
            r["arena"] = ir.ARENA_EXT0 + b.ext
            r["off"] = v.coff * b.esize
        return r


def compile_model(desc, weights, batch, height, width, fetch_cols=(0,), want_probs=True, store=None, reuse=True, hilo=False,
                  ragged=False, input_norm=None, fuse_preprocess=False, chain=None, tail2=None, se_lateral=None, fallbacks=None):
    """reuse=False gives every buffer its own workspace range (debugging: all intermediates stay readable).
    hilo=True stores every conv / depthwise / transposed-conv weight as an fp16 hi + lo pair (F_HILO): ~22-bit weights for
    twice the MFMA work — for nets whose boxes must track an fp32 reference closely.
    ragged=True (recognisers): `width` is the widest sample of the batch; the plan runs with a per-sample width table
    (Program.width_table) and every sample gets the values a batch of its own width would have produced, bit for bit.
    se_lateral=False / tail2=False: compile without that rewrite from the start; `fallbacks` (a dict): the rewrites this call had to
    abandon are recorded there as {"tail2": False, "se_lateral": False} so that the caller (engine.Net) passes them for the next shape
    instead of paying the failed attempt again."""
    snap = store.snapshot() if store is not None else None

    def build(se_lateral, tail2=tail2):
        if snap is not None:
            store.rollback(snap)         # blobs of an abandoned attempt leave the shared store (nothing refers to their offsets)
        c = Compiler(desc, weights, batch, height, width, fetch_cols, want_probs, store, reuse, se_lateral=se_lateral, tail2=tail2)
        c.ragged = bool(ragged)
        c.hilo = bool(hilo)
        c.input_norm = input_norm      # (mean3, std3): the plan takes RAW resized pixels + a ones channel (Compiler.fold_input_norm)
        c.fuse_preprocess = bool(fuse_preprocess)    # ... and resizes them itself from the uint8 frames (F_U8SRC): the plan input IS the frames
        c.chain = bool(hilo) if chain is None else bool(chain)      # 1x1 / depthwise chains as OP_CHAIN (chains.py): the hi + lo nets (mobile detectors)
        c.chain_lo = _dev_switch("VSE_CHAIN_LO", "1") != "0"     # tensors that feed a chain are stored as fp16 hi + lo pairs
        if c.hilo:
            c.use_patch = False          # conv_patch_kernel has no two-pass K walk (the implicit-GEMM, stem and column kernels do)
        return c.compile()
    try:
        try:
            return build(se_lateral)
        except Tail2Unsupported:
            # the dense map of an F_TAIL2 conv met a reader that cannot take it: the two transposed convs as separate launches
            tail2 = False
            if fallbacks is not None:
                fallbacks["tail2"] = False
            return build(se_lateral, False)
    except GatedConvUnsupported:
        # _rewrite_se_laterals turned a 1x1 conv + SE block into a gated conv whose surroundings the epilogue cannot express (an affine or
        # an activation behind the SE add, a gate shape, a kernel family): the graph compiled before that rewrite existed — compile it
        # without the rewrite instead of refusing it
        if fallbacks is not None:
            fallbacks["se_lateral"] = False
        return build(False, tail2)
