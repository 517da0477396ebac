"""ORACLE (test infrastructure, not product code) — CPU fp32 restatement of the det/rec network graphs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED: the reference (eritpchy/video-subtitle-extractor v2.2.0) executes these graphs inside
third-party `paddlepaddle==3.0.0` (README_en.md:186,195) via `paddleocr~=2.10.0` (requirements.txt:16),
called from backend/tools/ocr.py:27 and backend/tools/subtitle_detect.py:25.  Neither package nor any
golden output exists in the reference checkout, so this interpreter is anchored on the *graphs themselves*
(backend/models/**/inference.pdmodel, converted to JSON descriptors by tools/pdmodel_convert.py) and on
the published semantics of each Paddle operator.

The interpreter walks the descriptor op by op in NCHW / fp32 on torch-CPU, exactly one torch call per
Paddle op, no fusion, no layout change — deliberately the *dumbest* possible execution so that it shares
no structure with the HIP engine's compiler (fusion, BN folding, NHWC, fp16).

Consistency checks (NOT pins: neither paddle nor cv2 can run here): tests/test_oracle_crosschecks.py compares the restated
primitives of this file with independent implementations of the same published definitions (torch.nn.LSTM, a float64
bilinear resize, brute-force rotation search, scipy's convex hull).
"""
import json
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

MODELS_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                          "video-subtitle-extractor_amd", "models")


def load_descriptor(model_id):
    with open(os.path.join(MODELS_DIR, model_id + ".json")) as f:
        return json.load(f)


def load_real_weights(model_id):
    """Real weights exist only for V3_ch_det_fast."""
    p = os.path.join(MODELS_DIR, model_id + ".npz")
    if not os.path.exists(p):
        return None
    z = np.load(p)
    return {k: z[k] for k in z.files}


def synth_weights(desc, seed=0):
    """Seeded stand-in weights for graphs whose .pdiparams blob is missing from the reference checkout.

    He-normal convs/linears (gain chosen so activations stay O(1) through the depth), BN close to identity
    with mild per-channel spread so that BN folding is actually exercised, small biases.
    Deterministic in (model, seed): generated on the fly on both boxes, never stored.
    """
    rng = np.random.default_rng(seed)
    role = {}
    for op in desc["ops"]:
        t = op["type"]
        if t in ("conv2d", "depthwise_conv2d"):
            role[op["in"]["Filter"][0]] = "conv"
        elif t == "conv2d_transpose":
            role[op["in"]["Filter"][0]] = "convT"
        elif t == "batch_norm":
            role[op["in"]["Scale"][0]] = "bn_scale"
            role[op["in"]["Bias"][0]] = "bn_bias"
            role[op["in"]["Mean"][0]] = "bn_mean"
            role[op["in"]["Variance"][0]] = "bn_var"
        elif t == "layer_norm":
            role[op["in"]["Scale"][0]] = "ln_scale"
            role[op["in"]["Bias"][0]] = "ln_bias"
        elif t in ("matmul_v2", "matmul"):
            y = op["in"]["Y"][0]
            if y in desc["params"]:
                role[y] = "linear"
        elif t == "rnn":
            for w in op["in"]["WeightList"]:
                role[w] = "lstm"
        elif t in ("elementwise_add", "elementwise_mul"):
            # (the parameter is X in the PP-LCNetV3 "learnable affine" multiplies, Y everywhere else: until round 5 only Y was looked
            # at, the LAB scales fell to the default N(0, 0.05) and every LCNetV3 stand-in multiplied its signal by ~0.05 twice per
            # unit — after a dozen units the net was a CONSTANT function of its input)
            for y in (op["in"]["Y"][0], op["in"]["X"][0]):
                if y in desc["params"] and y not in role:
                    n = int(np.prod(desc["params"][y]["dims"]))
                    if n == 1:
                        role[y] = "lab_scale" if t == "elementwise_mul" else "lab_bias"
                    else:
                        role[y] = "bias"
    out = {}
    for name in sorted(desc["params"]):
        dims = desc["params"][name]["dims"]
        r = role.get(name, "bias")
        if r == "conv":
            fan_in = dims[1] * dims[2] * dims[3]
            a = rng.standard_normal(dims) * math.sqrt(2.0 / fan_in)
        elif r == "convT":
            fan_in = dims[0]  # 2x2 s2: each output pixel sees Cin taps once
            a = rng.standard_normal(dims) * math.sqrt(2.0 / fan_in)
        elif r == "linear":
            a = rng.standard_normal(dims) * math.sqrt(1.0 / dims[0])
        elif r == "lstm":
            a = rng.uniform(-1, 1, dims) * (1.0 / math.sqrt(256.0))
        elif r == "bn_scale":
            a = rng.uniform(0.8, 1.2, dims)
        elif r == "bn_var":
            a = rng.uniform(0.5, 1.5, dims)
        elif r == "bn_mean":
            a = rng.standard_normal(dims) * 0.1
        elif r == "bn_bias":
            a = rng.standard_normal(dims) * 0.1
        elif r == "ln_scale":
            a = rng.uniform(0.8, 1.2, dims)
        elif r == "ln_bias":
            a = rng.standard_normal(dims) * 0.05
        elif r == "lab_scale":
            a = rng.uniform(0.8, 1.25, dims)
        elif r == "lab_bias":
            a = rng.standard_normal(dims) * 0.05
        else:
            a = rng.standard_normal(dims) * 0.05
        out[name] = a.astype(np.float32)
    return out


_CALIB_SHAPES = {"det": (1, 3, 96, 160), "rec": (2, 3, 48, 160), "rec32": (2, 3, 32, 160)}


_CALIB_MIN_SAMPLES = 16        # per-channel statistics need this many values per channel (the SE convs see one pixel per image)
# Where a channel sits relative to its activation after the calibration pass: centred (batch mean removed) and then SHIFTED up by one
# standard deviation of its input-dependent part.  Measured on the stand-ins (tools/standin_study.py): without the shift every
# conv -> centre -> ReLU unit amplifies a perturbation by ~1.2 (the chaotic regime of batch-normalised random nets): fp16 WEIGHT rounding
# alone moved the recognisers' log-probabilities by 0.4-0.8; one sigma up (84 % of the units on the linear side of the ReLU) the same
# nets still answer different inputs differently (median |delta log p| between two inputs ~1.2) and weight rounding moves them by
# 6-8e-2 at most, 6-9e-3 in the median; two sigma starts to flatten them again.
_CALIB_CENTER = float(os.environ.get("VSE_CALIB_CENTER", "1.0"))
_CALIB_SHIFT = float(os.environ.get("VSE_CALIB_SHIFT", "1.0"))


def _calib_scale(y, ch_dim):
    """Per-OUTPUT-CHANNEL divisor of a conv / linear layer in the calibration pass: the standard deviation of the channel around
    ITS OWN mean over batch and positions — the part of the activation that depends on the input.  (Round 1-4 divided by the
    std over all elements, which per-channel offsets dominate: the input-dependent part shrank layer by layer until the deep
    stand-in nets answered every input alike.)  Channels with (almost) no variation, and layers with fewer than
    _CALIB_MIN_SAMPLES values per channel, fall back to the layer-wide figure."""
    n = y.shape[ch_dim]
    allsd = float(y.std()) if y.numel() > 1 else 1.0
    allsd = allsd if allsd > 0 else 1.0
    if y.numel() // n < _CALIB_MIN_SAMPLES:
        return torch.full((n,), allsd)
    red = [d for d in range(y.dim()) if d != ch_dim]
    sd = y.std(red, unbiased=False)
    floor = 0.1 * float(sd.median()) if float(sd.median()) > 0 else allsd
    return torch.where(sd > floor, sd, torch.full_like(sd, max(floor, 1e-12)))


def calibrate(desc, weights, seed=0):
    """Data-dependent rescale of the synthetic weights (LSUV-style, per channel): one forward pass on a fixed seeded input; every conv /
    linear output channel is divided by the standard deviation of ITS input-dependent part (_calib_scale), batch-norm statistics become
    the pass's batch statistics (with the seeded spread), per-channel biases behind convs centre their channel — every layer emits O(1)
    activations that still DEPEND ON THE INPUT (tests/test_oracle_crosschecks.py checks that: two inputs, different outputs), and the
    head is not saturated.  Deterministic given (descriptor, seed); cheap (one tiny forward)."""
    mid = desc["model"]
    kind = "det" if "_det" in mid else ("rec32" if mid.startswith("V2_") else "rec")
    x = np.random.default_rng(1000 + seed).uniform(-1, 1, _CALIB_SHAPES[kind]).astype(np.float32)
    run_graph(desc, weights, x, _calibrate=True)
    return weights


def get_weights(model_id, seed=0):
    """(descriptor, weights).  Real weights when the blob exists (V3_ch_det_fast), else calibrated stand-ins."""
    desc = load_descriptor(model_id)
    w = load_real_weights(model_id)
    if w is None:
        w = calibrate(desc, synth_weights(desc, seed), seed)
    return desc, w


# ------------------------------------------------------------------------------------------------ ops
def _bcast(x, y, axis):
    """Paddle elementwise broadcasting: y's dims are aligned to x starting at `axis` (-1 = trailing)."""
    if y.dim() == x.dim():
        return y
    if axis == -1:
        axis = x.dim() - y.dim()
    shape = [1] * axis + list(y.shape) + [1] * (x.dim() - axis - y.dim())
    return y.reshape(shape)


def _pad2(p):
    if len(p) == 2:
        return (p[0], p[1])
    assert p[0] == p[1] and p[2] == p[3], p
    return (p[0], p[2])


def _lstm_ref(x, weights, num_layers, bidirec, hidden):
    """Paddle `rnn` op, mode LSTM (WeightList = all (w_ih,w_hh) pairs layer-major /
    direction-minor, then all (b_ih,b_hh) pairs; gate order i,f,g,o).  x: [T,B,in]."""
    ndir = 2 if bidirec else 1
    ncell = num_layers * ndir
    ws = weights[:2 * ncell]
    bs = weights[2 * ncell:]
    inp = x
    for layer in range(num_layers):
        outs = []
        for d in range(ndir):
            c = layer * ndir + d
            w_ih, w_hh = ws[2 * c], ws[2 * c + 1]
            b_ih, b_hh = bs[2 * c], bs[2 * c + 1]
            T, B, _ = inp.shape
            h = torch.zeros(B, hidden)
            cst = torch.zeros(B, hidden)
            seq = range(T) if d == 0 else range(T - 1, -1, -1)
            hs = [None] * T
            for t in seq:
                g = inp[t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
                i, f, gg, o = g.chunk(4, dim=1)
                i, f, o = torch.sigmoid(i), torch.sigmoid(f), torch.sigmoid(o)
                gg = torch.tanh(gg)
                cst = f * cst + i * gg
                h = o * torch.tanh(cst)
                hs[t] = h
            outs.append(torch.stack(hs, 0))
        inp = torch.cat(outs, dim=2)
    return inp


def run_graph(desc, weights, x, return_all=False, _calibrate=False):
    """x: float32 NCHW tensor/ndarray.  Returns list of fetch outputs (by col) as torch tensors."""
    env = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}
    x = torch.as_tensor(x, dtype=torch.float32)
    fetch = {}
    # activations are dropped after their last reader (unless the caller wants them all): the 2176 x 3840 detector input of
    # the 4K stress configuration would otherwise keep ~100 GB of fp32 intermediates alive
    last = {}
    for oi, op in enumerate(desc["ops"]):
        for names in op["in"].values():
            for nm in names:
                last[nm] = oi
    dead = {}
    for nm, oi in last.items():
        if nm not in weights:
            dead.setdefault(oi + 1, []).append(nm)
    with torch.no_grad():
        for oi, op in enumerate(desc["ops"]):
            if not return_all:
                for nm in dead.get(oi, ()):
                    env.pop(nm, None)
            t = op["type"]
            a = op["attrs"]
            I = {k: [env[n] for n in v] for k, v in op["in"].items()} if t != "feed" else {}
            if t == "feed":
                env[op["out"]["Out"][0]] = x
                continue
            if t == "fetch":
                fetch[a.get("col", 0)] = I["X"][0]
                continue
            if t in ("conv2d", "depthwise_conv2d"):
                y = F.conv2d(I["Input"][0], I["Filter"][0], None, tuple(a["strides"]), _pad2(a["paddings"]),
                             tuple(a.get("dilations", [1, 1])), a.get("groups", 1))
                if _calibrate:
                    fn = op["in"]["Filter"][0]
                    sd = _calib_scale(y, 1)
                    weights[fn] = (weights[fn] / sd.numpy().reshape(-1, 1, 1, 1)).astype(np.float32)
                    y = y / sd.reshape(1, -1, 1, 1)
                env[op["out"]["Output"][0]] = y
            elif t == "conv2d_transpose":
                y = F.conv_transpose2d(I["Input"][0], I["Filter"][0], None, tuple(a["strides"]),
                                       _pad2(a["paddings"]), 0, a.get("groups", 1))
                env[op["out"]["Output"][0]] = y
            elif t == "batch_norm":
                xx = I["X"][0]
                sh = [1, -1] + [1] * (xx.dim() - 2)
                if _calibrate and xx.numel() // xx.shape[1] >= _CALIB_MIN_SAMPLES:
                    # running statistics as a trained net would hold them: the batch statistics of the calibration pass, with the
                    # seeded spread of the synthetic ones (mean + 0.1 N(0, 1) sigma, variance x U(0.5, 1.5)) so that folding is exercised
                    red = [d for d in range(xx.dim()) if d != 1]
                    m, v = xx.mean(red), xx.var(red, unbiased=False).clamp_min(1e-6)
                    mn, vn = op["in"]["Mean"][0], op["in"]["Variance"][0]
                    weights[mn] = (_CALIB_CENTER * m + (torch.from_numpy(weights[mn]) - _CALIB_SHIFT) * v.sqrt()).numpy().astype(np.float32)
                    weights[vn] = (v * torch.from_numpy(weights[vn])).numpy().astype(np.float32)
                    I["Mean"][0], I["Variance"][0] = torch.from_numpy(weights[mn]), torch.from_numpy(weights[vn])
                y = (xx - I["Mean"][0].reshape(sh)) / torch.sqrt(I["Variance"][0].reshape(sh) + a["epsilon"])
                y = y * I["Scale"][0].reshape(sh) + I["Bias"][0].reshape(sh)
                env[op["out"]["Y"][0]] = y
            elif t == "pool2d":
                xx = I["X"][0]
                if a.get("adaptive", False):
                    assert a["pooling_type"] == "avg"
                    y = F.adaptive_avg_pool2d(xx, tuple(a["ksize"]))
                elif a.get("global_pooling", False):
                    y = xx.mean((2, 3), keepdim=True) if a["pooling_type"] == "avg" else xx.amax((2, 3), keepdim=True)
                elif a["pooling_type"] == "max":
                    y = F.max_pool2d(xx, tuple(a["ksize"]), tuple(a["strides"]), _pad2(a["paddings"]),
                                     ceil_mode=a.get("ceil_mode", False))
                else:
                    y = F.avg_pool2d(xx, tuple(a["ksize"]), tuple(a["strides"]), _pad2(a["paddings"]),
                                     ceil_mode=a.get("ceil_mode", False),
                                     count_include_pad=not a.get("exclusive", True))
                env[op["out"]["Out"][0]] = y
            elif t == "relu":
                env[op["out"]["Out"][0]] = torch.relu(I["X"][0])
            elif t == "sigmoid":
                env[op["out"]["Out"][0]] = torch.sigmoid(I["X"][0])
            elif t == "hard_swish":
                xx = I["X"][0]
                env[op["out"]["Out"][0]] = xx * torch.clamp(xx + a["offset"], 0.0, a["threshold"]) / a["scale"]
            elif t == "hard_sigmoid":
                env[op["out"]["Out"][0]] = torch.clamp(I["X"][0] * a["slope"] + a["offset"], 0.0, 1.0)
            elif t == "swish":
                xx = I["X"][0]
                env[op["out"]["Out"][0]] = xx * torch.sigmoid(a.get("beta", 1.0) * xx)
            elif t == "elementwise_add":
                xx, yy = I["X"][0], I["Y"][0]
                yn = op["in"]["Y"][0]
                if (_calibrate and yn in weights and xx.dim() == 4 and yy.dim() == 1 and yy.numel() == xx.shape[1] > 1
                        and a.get("axis", -1) == 1 and xx.numel() // xx.shape[1] >= _CALIB_MIN_SAMPLES):
                    # a per-channel bias behind a conv (the re-parameterised nets have no batch norm): centre the channel like the BN it
                    # stands for would have, keep the seeded bias as the spread around it
                    weights[yn] = (torch.from_numpy(weights[yn]) - _CALIB_CENTER * xx.mean((0, 2, 3))
                                   + _CALIB_SHIFT * xx.std((0, 2, 3), unbiased=False)).numpy().astype(np.float32)
                    yy = torch.from_numpy(weights[yn])
                env[op["out"]["Out"][0]] = xx + _bcast(xx, yy, a.get("axis", -1))
            elif t == "elementwise_mul":
                xx, yy = I["X"][0], I["Y"][0]
                env[op["out"]["Out"][0]] = xx * _bcast(xx, yy, a.get("axis", -1))
            elif t == "nearest_interp_v2":
                s = a["scale"]
                assert not a.get("align_corners", False)
                env[op["out"]["Out"][0]] = F.interpolate(I["X"][0], scale_factor=(s[0], s[1]), mode="nearest")
            elif t == "layer_norm":
                xx = I["X"][0]
                ax = a["begin_norm_axis"]
                env[op["out"]["Y"][0]] = F.layer_norm(xx, xx.shape[ax:], I["Scale"][0].reshape(xx.shape[ax:]),
                                                      I["Bias"][0].reshape(xx.shape[ax:]), a["epsilon"])
            elif t == "softmax":
                env[op["out"]["Out"][0]] = torch.softmax(I["X"][0], dim=a["axis"])
            elif t == "scale":
                xx = I["X"][0]
                if a.get("bias_after_scale", True):
                    y = xx * a["scale"] + a.get("bias", 0.0)
                else:
                    y = (xx + a.get("bias", 0.0)) * a["scale"]
                env[op["out"]["Out"][0]] = y.to(xx.dtype) if xx.dtype.is_floating_point else y
            elif t == "matmul_v2":
                xx, yy = I["X"][0], I["Y"][0]
                if a.get("trans_x", False):
                    xx = xx.transpose(-1, -2)
                if a.get("trans_y", False):
                    yy = yy.transpose(-1, -2)
                y = torch.matmul(xx, yy)
                yn = op["in"]["Y"][0]
                if _calibrate and yn in weights:
                    sd = _calib_scale(y, y.dim() - 1)
                    weights[yn] = (weights[yn] / sd.numpy().reshape(1, -1)).astype(np.float32)
                    y = y / sd
                env[op["out"]["Out"][0]] = y
            elif t == "matmul":
                xx, yy = I["X"][0], I["Y"][0]
                if a.get("transpose_X", False):
                    xx = xx.transpose(-1, -2)
                if a.get("transpose_Y", False):
                    yy = yy.transpose(-1, -2)
                env[op["out"]["Out"][0]] = torch.matmul(xx, yy) * a.get("alpha", 1.0)
            elif t == "transpose2":
                env[op["out"]["Out"][0]] = I["X"][0].permute(*a["axis"]).contiguous()
            elif t == "reshape2":
                xx = I["X"][0]
                if "ShapeTensor" in I and I["ShapeTensor"]:
                    shape = [int(s.reshape(-1)[0]) for s in I["ShapeTensor"]]
                else:
                    shape = list(a["shape"])
                shape = [xx.shape[i] if s == 0 else s for i, s in enumerate(shape)]
                env[op["out"]["Out"][0]] = xx.reshape(shape)
            elif t == "slice":
                xx = I["Input"][0]
                idx = [slice(None)] * xx.dim()
                for ax, s, e in zip(a["axes"], a["starts"], a["ends"]):
                    idx[ax] = slice(s, min(e, xx.shape[ax]))
                y = xx[tuple(idx)]
                for ax in sorted(a.get("decrease_axis", []), reverse=True):
                    y = y.squeeze(ax)
                env[op["out"]["Out"][0]] = y
            elif t == "concat":
                env[op["out"]["Out"][0]] = torch.cat(I["X"], dim=a["axis"])
            elif t == "squeeze2":
                y = I["X"][0]
                for ax in sorted(a["axes"], reverse=True):
                    y = y.squeeze(ax)
                env[op["out"]["Out"][0]] = y
            elif t == "flatten_contiguous_range":
                env[op["out"]["Out"][0]] = I["X"][0].flatten(a["start_axis"], a["stop_axis"])
            elif t == "dropout":
                assert a.get("dropout_implementation") == "upscale_in_train"
                env[op["out"]["Out"][0]] = I["X"][0]
            elif t == "assign":
                env[op["out"]["Out"][0]] = I["X"][0]
            elif t == "shape":
                env[op["out"]["Out"][0]] = torch.tensor(list(I["Input"][0].shape), dtype=torch.int32)
            elif t == "fill_constant":
                dt = {2: torch.int32, 3: torch.int64, 5: torch.float32}[a.get("dtype", 5)]
                val = a.get("value", 0.0)
                if a.get("str_value"):
                    val = float(a["str_value"])
                env[op["out"]["Out"][0]] = torch.full(list(a["shape"]), val).to(dt)
            elif t == "fill_constant_batch_size_like":
                ref = I["Input"][0]
                shape = list(a["shape"])
                shape[a.get("output_dim_idx", 0)] = ref.shape[a.get("input_dim_idx", 0)]
                env[op["out"]["Out"][0]] = torch.full(shape, float(a.get("value", 0.0)))
            elif t == "rnn":
                assert a["mode"] == "LSTM"
                y = _lstm_ref(I["Input"][0], I["WeightList"], a["num_layers"], a["is_bidirec"], a["hidden_size"])
                env[op["out"]["Out"][0]] = y
            else:
                raise NotImplementedError(t)
    outs = [fetch[c] for c in sorted(fetch)]
    if return_all:
        return outs, env
    return outs
