#!/usr/bin/env python3
"""Randomised differential test of the graph COMPILER: random small CNN graphs over the operator set of the reference's models
(conv / depthwise / transposed conv, batch norm, bias, activations, pooling, global pooling + squeeze-excite gates, residual adds,
concats, nearest x2 upsampling + lateral adds, scale) are compiled and run
  * through the CPU emulator of the engine program (oracle/ir_emul.py) — default, needs no GPU: checks lowering, fusion, buffer
    reuse and every weight packing, or
  * through the HIP engine (--gpu): checks the kernels on the same programs,
against the op-by-op fp32 interpreter (oracle/net_ref.py).  A graph the compiler refuses with UnsupportedGraph counts as
"refused" (loud, allowed); any other exception or a numeric mismatch is a failure.
usage: python tools/fuzz_graph.py [--cases 200] [--seed 0] [--gpu]"""
import argparse
import os
import sys
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import ir_emul, net_ref
from vse_amd import compiler


class G:
    def __init__(self, rng):
        self.rng, self.ops, self.params, self.w, self.shapes = rng, [], {}, {}, {}
        self.k = 0
        self.tensors = []                    # (name, channels, log2 stride)
        self.ops.append({"type": "feed", "in": {"X": ["feed"]}, "out": {"Out": ["x"]}, "attrs": {"col": 0}})

    def name(self, p):
        self.k += 1
        return f"{p}_{self.k}"

    def param(self, name, arr):
        self.params[name] = {"dims": list(arr.shape), "dtype": 5}
        self.w[name] = arr.astype(np.float32)

    def conv(self, src, cin, cout, k, s, p, groups=1, typ="conv2d"):
        out, wn = self.name("conv"), self.name("w")
        fan = cin // groups * k[0] * k[1]
        self.param(wn, self.rng.standard_normal((cout, cin // groups, k[0], k[1])) / np.sqrt(fan))
        self.ops.append({"type": typ, "in": {"Input": [src], "Filter": [wn]}, "out": {"Output": [out]},
                         "attrs": {"strides": list(s), "paddings": list(p), "dilations": [1, 1], "groups": groups,
                                   "padding_algorithm": "EXPLICIT", "data_format": "NCHW"}})
        self.shapes[out] = [-1, cout, -1, -1]
        return out

    def deconv(self, src, cin, cout):
        out, wn = self.name("deconv"), self.name("w")
        self.param(wn, self.rng.standard_normal((cin, cout, 2, 2)) / np.sqrt(cin))
        self.ops.append({"type": "conv2d_transpose", "in": {"Input": [src], "Filter": [wn]}, "out": {"Output": [out]},
                         "attrs": {"strides": [2, 2], "paddings": [0, 0], "dilations": [1, 1], "groups": 1, "padding_algorithm": "EXPLICIT",
                                   "output_padding": [], "output_size": [], "data_format": "NCHW"}})
        self.shapes[out] = [-1, cout, -1, -1]
        return out

    def bn(self, src, c):
        out = self.name("bn")
        names = {k: self.name("bn" + k) for k in ("Scale", "Bias", "Mean", "Variance")}
        self.param(names["Scale"], self.rng.uniform(0.7, 1.3, c))
        self.param(names["Bias"], self.rng.uniform(-0.3, 0.3, c))
        self.param(names["Mean"], self.rng.uniform(-0.2, 0.2, c))
        self.param(names["Variance"], self.rng.uniform(0.6, 1.4, c))
        self.ops.append({"type": "batch_norm", "in": {"X": [src], **{k: [v] for k, v in names.items()}},
                         "out": {"Y": [out], "MeanOut": [names["Mean"]], "VarianceOut": [names["Variance"]],
                                 "SavedMean": [self.name("sm")], "SavedVariance": [self.name("sv")]},
                         "attrs": {"epsilon": 1e-5, "data_layout": "NCHW"}})
        self.shapes[out] = [-1, c, -1, -1]
        return out

    def bias(self, src, c):
        out, bn_ = self.name("biased"), self.name("b")
        self.param(bn_, self.rng.uniform(-0.3, 0.3, c))
        self.ops.append({"type": "elementwise_add", "in": {"X": [src], "Y": [bn_]}, "out": {"Out": [out]}, "attrs": {"axis": 1}})
        self.shapes[out] = [-1, c, -1, -1]
        return out

    def act(self, src, c, kind):
        out = self.name(kind)
        attrs = {"hard_swish": {"offset": 3.0, "scale": 6.0, "threshold": 6.0}, "hard_sigmoid": {"slope": 0.2, "offset": 0.5},
                 "swish": {"beta": 1.0}}.get(kind, {})
        self.ops.append({"type": kind, "in": {"X": [src]}, "out": {"Out": [out]}, "attrs": attrs})
        self.shapes[out] = [-1, c, -1, -1]
        return out

    def binary(self, a, b, c, typ="elementwise_add", axis=-1):
        out = self.name("add" if typ == "elementwise_add" else "mul")
        self.ops.append({"type": typ, "in": {"X": [a], "Y": [b]}, "out": {"Out": [out]}, "attrs": {"axis": axis}})
        self.shapes[out] = [-1, c, -1, -1]
        return out

    def pool(self, src, c, kind, k, s, p, glob=False):
        out = self.name("pool")
        self.ops.append({"type": "pool2d", "in": {"X": [src]}, "out": {"Out": [out]},
                         "attrs": {"pooling_type": kind, "ksize": [k, k], "strides": [s, s], "paddings": [p, p], "ceil_mode": False,
                                   "exclusive": True, "adaptive": bool(glob), "global_pooling": False, "padding_algorithm": "EXPLICIT"}})
        self.shapes[out] = [-1, c, -1, -1]
        return out

    def up2(self, src, c):
        out = self.name("up")
        self.ops.append({"type": "nearest_interp_v2", "in": {"X": [src]}, "out": {"Out": [out]},
                         "attrs": {"scale": [2.0, 2.0], "out_h": -1, "out_w": -1, "align_corners": False, "interp_method": "nearest",
                                   "data_layout": "NCHW"}})
        self.shapes[out] = [-1, c, -1, -1]
        return out

    def concat(self, srcs, c):
        out = self.name("cat")
        self.ops.append({"type": "concat", "in": {"X": list(srcs)}, "out": {"Out": [out]}, "attrs": {"axis": 1}})
        self.shapes[out] = [-1, c, -1, -1]
        return out

    def scale(self, src, c):
        out = self.name("scale")
        self.ops.append({"type": "scale", "in": {"X": [src]}, "out": {"Out": [out]},
                         "attrs": {"scale": float(self.rng.choice([0.5, 2.0, 1.0])), "bias": float(self.rng.choice([0.0, 0.1])), "bias_after_scale": True}})
        self.shapes[out] = [-1, c, -1, -1]
        return out

    def finish(self, src):
        self.ops.append({"type": "fetch", "in": {"X": [src]}, "out": {"Out": ["fetch"]}, "attrs": {"col": 0}})
        return {"model": "fuzz", "ops": self.ops, "params": self.params, "var_shapes": self.shapes}, self.w


def random_graph(rng, h, w):
    g = G(rng)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    acts = ["relu", "hard_swish", "sigmoid", "hard_sigmoid", "swish", None, "relu"]
    cur, c, ls = "x", 3, 0
    hist = []                                   # (name, c, log2 stride)

    def hw():
        return h >> ls, w >> ls

    def conv_block(cin, cout, k, s):
        nonlocal cur
        p = (k[0] // 2, k[1] // 2)
        t = g.conv(cur, cin, cout, k, (s, s), p)
        r = rng.random()
        if r < 0.6:
            t = g.bn(t, cout)
        elif r < 0.85:
            t = g.bias(t, cout)
        a = pick(acts)
        if a:
            t = g.act(t, cout, a)
        cur = t
    # stem like every model of the reference: 3x3 stride 2 from 3 channels
    c0 = pick([8, 16, 24, 32, 48])
    conv_block(3, c0, (3, 3), 2)
    c, ls = c0, 1
    hist.append((cur, c, ls))
    for _ in range(int(rng.integers(3, 12))):
        kind = pick(["conv", "conv", "conv", "conv1", "dw", "pool", "res", "cat", "se", "sefold", "fpn", "deconv", "scale", "conv9"])
        hh, ww = hw()
        if kind == "conv":
            k = pick([(3, 3), (3, 3), (5, 5), (1, 3), (3, 1)])
            s = 2 if (rng.random() < 0.25 and min(hh, ww) >= 8) else 1
            cout = pick([8, 16, 24, 32, 40, 64, 72, 96, 128, 12, 20, 36, 100])
            conv_block(c, cout, k, s)
            c, ls = cout, ls + (s == 2)
        elif kind == "conv1":
            cout = pick([8, 16, 24, 32, 64, 96, 160, 6, 30, 50])
            conv_block(c, cout, (1, 1), 1)
            c = cout
        elif kind == "conv9" and min(hh, ww) >= 8 and c % 16 == 0:
            k = pick([(9, 9), (7, 7), (5, 5)])
            cout = pick([16, 32, 64])
            conv_block(c, cout, k, 1)
            c = cout
        elif kind == "dw":
            k = pick([3, 5])
            s = 2 if (rng.random() < 0.3 and min(hh, ww) >= 8) else 1
            t = g.conv(cur, c, c, (k, k), (s, s), (k // 2, k // 2), groups=c, typ="depthwise_conv2d")
            t = g.bn(t, c)
            a = pick(acts)
            cur = g.act(t, c, a) if a else t
            ls += (s == 2)
        elif kind == "pool" and min(hh, ww) >= 8:
            typ, k, s, p = pick([("max", 3, 2, 1), ("max", 2, 2, 0), ("avg", 2, 2, 0), ("avg", 3, 2, 1), ("max", 3, 1, 1)])
            cur = g.pool(cur, c, typ, k, s, p)
            ls += (s == 2)
        elif kind == "res":
            same = [t for t in hist if t[1] == c and t[2] == ls and t[0] != cur]
            if same:
                cur = g.binary(cur, pick(same)[0], c)
                if rng.random() < 0.5:
                    cur = g.act(cur, c, "relu")
        elif kind == "cat":
            same = [t for t in hist if t[2] == ls and t[0] != cur]
            if same:
                others = [same[int(i)] for i in rng.permutation(len(same))[:int(rng.integers(1, 3))]]
                cur = g.concat([cur] + [t[0] for t in others], c + sum(t[1] for t in others))
                c += sum(t[1] for t in others)
        elif kind == "se" and c >= 8:
            gp = g.pool(cur, c, "avg", 1, 1, 0, glob=True)
            mid = max(8, c // 4)
            t = g.act(g.bias(g.conv(gp, c, mid, (1, 1), (1, 1), (0, 0)), mid), mid, "relu")
            t = g.act(g.bias(g.conv(t, mid, c, (1, 1), (1, 1), (0, 0)), c), c, pick(["hard_sigmoid", "sigmoid"]))
            cur = g.binary(cur, t, c, "elementwise_mul", axis=-1)
        elif kind == "sefold" and hh * ww >= 256:
            # SE output read only by a depthwise conv and a 1x1 conv: the compiler folds the gate into both (F_GATE + F_IMGW)
            cse = pick([128, 192, 256])
            conv_block(c, cse, (1, 1), 1)
            c = cse
            gp = g.pool(cur, c, "avg", 1, 1, 0, glob=True)
            mid = max(8, c // 4)
            t = g.act(g.bias(g.conv(gp, c, mid, (1, 1), (1, 1), (0, 0)), mid), mid, "relu")
            t = g.act(g.bias(g.conv(t, mid, c, (1, 1), (1, 1), (0, 0)), c), c, pick(["hard_sigmoid", "sigmoid"]))
            se = g.binary(cur, t, c, "elementwise_mul", axis=-1)
            lat = g.act(g.bias(g.conv(se, c, c, (1, 1), (1, 1), (0, 0)), c), c, "relu")
            dw = g.bn(g.conv(se, c, c, (3, 3), (1, 1), (1, 1), groups=c, typ="depthwise_conv2d"), c)
            cur = g.binary(lat, dw, c)
        elif kind == "fpn" and ls >= 2:
            big = [t for t in hist if t[2] == ls - 1]
            if big:
                lat = pick(big)
                cur_keep = cur
                cur = lat[0]
                save_c = c
                t_lat = g.conv(lat[0], lat[1], c, (1, 1), (1, 1), (0, 0))
                cur = g.binary(t_lat, g.up2(cur_keep, save_c), c)
                ls -= 1
        elif kind == "deconv" and ls >= 1 and min(hh, ww) <= 64:
            cout = pick([8, 16, 32, 64])
            t = g.deconv(cur, c, cout)
            cur = g.act(g.bn(t, cout), cout, "relu")
            c, ls = cout, ls - 1
        elif kind == "scale":
            cur = g.scale(cur, c)
        hist.append((cur, c, ls))
    if rng.random() < 0.5:
        cout = pick([1, 8, 16])
        cur = g.act(g.bias(g.conv(cur, c, cout, (1, 1), (1, 1), (0, 0)), cout), cout, "sigmoid")
        c = cout
    return g.finish(cur) + (c,)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--hilo", action="store_true", help="fp16 hi + lo weight pairs (the mobile detectors' default)")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    if a.gpu:
        import torch
        from vse_amd import engine
        ctx = engine.Context(0)
    bad, refused, ok = [], 0, 0
    for i in range(a.cases):
        h, w = int(rng.integers(2, 9)) * 16, int(rng.integers(2, 11)) * 16
        n = int(rng.integers(1, 4))
        desc, wts, cout = random_graph(rng, h, w)
        x = rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float16).astype(np.float32)
        summary = [op["type"] for op in desc["ops"]]
        try:
            ref = net_ref.run_graph(desc, wts, x)[0].numpy()
        except Exception as e:                                       # noqa: BLE001 - the generator made an invalid graph
            bad.append((i, "oracle failed: " + type(e).__name__ + str(e)[:100], summary))
            continue
        try:
            if a.gpu:
                net = engine.Net(ctx, desc, wts, want_probs=True, hilo=a.hilo)
                xt = torch.from_numpy(ir_emul.to_nhwc8(x).astype(np.float16)).to(ctx.tdev)
                got = net.run(xt)[0].float().cpu().numpy()
            else:
                prog = compiler.compile_model(desc, wts, n, h, w, hilo=a.hilo)
                got = ir_emul.Emulator(prog).run(ir_emul.to_nhwc8(x))[0]
        except compiler.UnsupportedGraph:
            refused += 1
            continue
        except Exception as e:                                       # noqa: BLE001
            bad.append((i, type(e).__name__ + ": " + str(e)[:160], summary, traceback.format_exc().splitlines()[-3:]))
            continue
        got = np.transpose(np.asarray(got)[..., :cout], (0, 3, 1, 2))
        if got.shape != ref.shape:
            bad.append((i, f"shape {got.shape} vs {ref.shape}", summary))
            continue
        err = np.abs(got - ref).max()
        tol = (2e-2 if a.gpu else 5e-3) * max(1.0, np.abs(ref).max())
        if not np.isfinite(got).all() or err > tol:
            bad.append((i, f"max err {err:.4g} (tol {tol:.3g}, ref max {np.abs(ref).max():.3g})", (n, h, w), summary))
        else:
            ok += 1
    print(f"{a.cases} graphs: {ok} match, {refused} refused (UnsupportedGraph), {len(bad)} failures")
    for b in bad[:25]:
        print("FAIL", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
