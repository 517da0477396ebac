#!/usr/bin/env python3
"""Single-layer timing of the conv kernels on the GPU, per conv_gemm_kernel tile configuration.

usage: python tools/bench_conv.py [--cfgs 0,3,4,5,6,8] [--layers name,name...]
Each layer is a two-op graph (1x1 stem lifting the 3-channel feed to `cin`, then the conv under test); the conv under
test is timed with HIP events (vse_plan_profile), min of 3.  VSE_GEMM_CFG is read by the launcher at every launch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vse_amd import engine

LAYERS = {  # name: cin, cout, k, stride, pad, h, w, n
    "det_1x1_896_256": (896, 256, (1, 1), (1, 1), (0, 0), 136, 240, 64),
    "det_1x1_1216_512": (1216, 512, (1, 1), (1, 1), (0, 0), 68, 120, 64),
    "det_1x1_256_256": (256, 256, (1, 1), (1, 1), (0, 0), 136, 240, 64),
    "det_1x1_1920_768": (1920, 768, (1, 1), (1, 1), (0, 0), 34, 60, 64),
    "det_1x1_512_256": (512, 256, (1, 1), (1, 1), (0, 0), 68, 120, 64),
    "det_1x1_768_256": (768, 256, (1, 1), (1, 1), (0, 0), 34, 60, 64),
    "det_3x3_64_128": (64, 128, (3, 3), (1, 1), (1, 1), 272, 480, 64),
    "det_3x3_64_64": (64, 64, (3, 3), (1, 1), (1, 1), 272, 480, 64),
    "det_1x1_64_256": (64, 256, (1, 1), (1, 1), (0, 0), 272, 480, 64),
    "det_1x1_896_256": (896, 256, (1, 1), (1, 1), (0, 0), 136, 240, 64),
    "det_1x1_1216_512": (1216, 512, (1, 1), (1, 1), (0, 0), 68, 120, 64),
    "det_1x1_256_256": (256, 256, (1, 1), (1, 1), (0, 0), 136, 240, 64),
    "det_1x1_1920_768": (1920, 768, (1, 1), (1, 1), (0, 0), 34, 60, 64),
    "rec_1x1_1216_512": (1216, 512, (1, 1), (1, 1), (0, 0), 12, 256, 32),
    "rec_1x1_1920_768": (1920, 768, (1, 1), (1, 1), (0, 0), 6, 256, 32),
    "rec_1x1_896_256": (896, 256, (1, 1), (1, 1), (0, 0), 12, 512, 32),
    "rec_3x3_224_224": (224, 224, (3, 3), (1, 1), (1, 1), 3, 256, 32),
    "det_3x3_192_192": (192, 192, (3, 3), (1, 1), (1, 1), 34, 60, 64),
    "det_3x3_160_160": (160, 160, (3, 3), (1, 1), (1, 1), 68, 120, 64),
    "rec_3x3_192_192": (192, 192, (3, 3), (1, 1), (1, 1), 6, 256, 32),
    "rec_3x3_160_160": (160, 160, (3, 3), (1, 1), (1, 1), 12, 256, 32),
    "det_3x3_128_128": (128, 128, (3, 3), (1, 1), (1, 1), 136, 240, 64),
    "det_3x3_256_64": (256, 64, (3, 3), (1, 1), (1, 1), 136, 240, 64),
    "det_3x3_256_160": (256, 160, (3, 3), (1, 1), (1, 1), 68, 120, 64),
    "det_3x3_768_192": (768, 192, (3, 3), (1, 1), (1, 1), 34, 60, 64),
    "det_9x9_256_64": (256, 64, (9, 9), (1, 1), (4, 4), 136, 240, 64),
    "det_9x9_64_64": (64, 64, (9, 9), (1, 1), (4, 4), 136, 240, 64),
    "det_9x9_256_64_h68": (256, 64, (9, 9), (1, 1), (4, 4), 68, 120, 64),
    "det_9x9_256_64_h34": (256, 64, (9, 9), (1, 1), (4, 4), 34, 60, 64),
    "det_9x9_256_64_h17": (256, 64, (9, 9), (1, 1), (4, 4), 17, 30, 64),
    "det_9x9_64_64_h68": (64, 64, (9, 9), (1, 1), (4, 4), 68, 120, 64),
    "det_3x3_256_64_h68": (256, 64, (3, 3), (1, 1), (1, 1), 68, 120, 64),
    "det_3x3_224_224": (224, 224, (3, 3), (1, 1), (1, 1), 17, 30, 64),
    "det_3x3_32_32": (32, 32, (3, 3), (1, 1), (1, 1), 136, 240, 64),
    "det_3x3_32_32_h68": (32, 32, (3, 3), (1, 1), (1, 1), 68, 120, 64),
    "det_1x1_32_64": (32, 64, (1, 1), (1, 1), (0, 0), 136, 240, 64),
    "det_1x1_64_32": (64, 32, (1, 1), (1, 1), (0, 0), 136, 240, 64),
    "rec_3x3_128_128_w768": (128, 128, (3, 3), (1, 1), (1, 1), 12, 384, 28),
    "rec_3x3_160_160_w768": (160, 160, (3, 3), (1, 1), (1, 1), 12, 192, 28),
    "rec_3x3_192_192_w768": (192, 192, (3, 3), (1, 1), (1, 1), 6, 192, 28),
    "rec_3x3_768_192_w768": (768, 192, (3, 3), (1, 1), (1, 1), 6, 192, 28),
    "rec_3x3_224_224_w768": (224, 224, (3, 3), (1, 1), (1, 1), 3, 192, 28),
    "det_7x7_32_32": (32, 32, (7, 7), (1, 1), (3, 3), 136, 240, 64),
    "det_5x5_32_32": (32, 32, (5, 5), (1, 1), (2, 2), 136, 240, 64),
    "rec_3x3_128_128": (128, 128, (3, 3), (1, 1), (1, 1), 12, 512, 32),
    "rec_3x3_768_192": (768, 192, (3, 3), (1, 1), (1, 1), 6, 256, 32),
    "rec_3x3_256_160": (256, 160, (3, 3), (1, 1), (1, 1), 12, 256, 32),
    "rec_3x3_64_128": (64, 128, (3, 3), (1, 1), (1, 1), 24, 512, 32),
}


def graph(cin, cout, k, s, p):
    desc = {"model": "unit", "ops": [
        {"type": "feed", "in": {"X": ["feed"]}, "out": {"Out": ["x"]}, "attrs": {"col": 0}},
        {"type": "conv2d", "in": {"Input": ["x"], "Filter": ["w0"]}, "out": {"Output": ["t0"]},
         "attrs": {"strides": [1, 1], "paddings": [0, 0], "groups": 1}},
        {"type": "conv2d", "in": {"Input": ["t0"], "Filter": ["w1"]}, "out": {"Output": ["t1"]},
         "attrs": {"strides": list(s), "paddings": list(p), "groups": 1}},
        {"type": "elementwise_add", "in": {"X": ["t1"], "Y": ["b1"]}, "out": {"Out": ["t2"]}, "attrs": {"axis": 1}},
        {"type": "relu", "in": {"X": ["t2"]}, "out": {"Out": ["t3"]}, "attrs": {}},
        {"type": "fetch", "in": {"X": ["t3"]}, "out": {"Out": ["fetch"]}, "attrs": {"col": 0}}],
        "params": {"w0": {"dims": [cin, 3, 1, 1], "dtype": 5}, "w1": {"dims": [cout, cin, k[0], k[1]], "dtype": 5},
                   "b1": {"dims": [cout], "dtype": 5}},
        "var_shapes": {"t0": [-1, cin, -1, -1], "t1": [-1, cout, -1, -1]}}
    rng = np.random.default_rng(1)
    wts = {"w0": rng.standard_normal((cin, 3, 1, 1)).astype(np.float32),
           "w1": (rng.standard_normal((cout, cin, k[0], k[1])) / np.sqrt(cin * k[0] * k[1])).astype(np.float32),
           "b1": rng.standard_normal(cout).astype(np.float32) * 0.1}
    return desc, wts


def main():
    cfgs = [c for c in (sys.argv[sys.argv.index("--cfgs") + 1].split(",") if "--cfgs" in sys.argv else ["0"])]
    names = sys.argv[sys.argv.index("--layers") + 1].split(",") if "--layers" in sys.argv else list(LAYERS)
    ctx = engine.Context(0)
    for name in names:
        cin, cout, k, s, p, h, w, n = LAYERS[name]
        desc, wts = graph(cin, cout, k, s, p)
        from vse_amd import compiler
        nets = {}
        # "p": conv_patch_kernel / conv_col_kernel allowed; "c": + conv_c3_kernel for any couts; numeric cfgs: implicit GEMM only
        for key, mink in (("g", 1 << 30), ("p", 500), ("c", 100)):
            compiler.PATCH_MIN_K = mink
            nets[key] = engine.Net(ctx, desc, wts, want_probs=False)
        x = (torch.rand((n, h, w, 8), device="cuda") * 2 - 1).half()
        x[..., 3:] = 0
        oh = (h + 2 * p[0] - k[0]) // s[0] + 1
        ow = (w + 2 * p[1] - k[1]) // s[1] + 1
        flops = 2.0 * n * oh * ow * cout * cin * k[0] * k[1]
        row = []
        ref = None
        for c in cfgs:
            net = nets[c if c in ("p", "c") else "g"]
            compiler.PATCH_MIN_K = (100 if c == "c" else 500) if c in ("p", "c") else 1 << 30      # plans are compiled lazily on the first run
            compiler.PATCH_MAX_COUT = 256 if c == "p" else 64       # "p": also try the patch kernel on wide layers (two+ cout tiles)
            compiler.COL3 = c == "c"
            compiler.COL3_MAX_COUT, compiler.COL3_MIN_TILE_EFF, compiler.COL3_WIDE_MIN_CIN = 4096, 0.0, 0
            compiler.COL3_MIN_K = 100
            os.environ["VSE_GEMM_CFG"] = "" if c in ("p", "d") else c      # "d": the launcher's own choice
            out = net.run(x)
            torch.cuda.synchronize()
            o = out[0].float()
            if ref is None:
                ref = o.clone()
            same = bool(torch.equal(o, ref)) or c in ("p", "c")
            best = 1e9
            for _ in range(3):
                ms, prog, var = net.profile(x)
                best = min(best, float(ms[1]))
            row.append(f"cfg{c}: {best:7.3f} ms {flops / best / 1e9:6.0f} TF/s{'' if same else ' MISMATCH'}")
        print(f"{name:20s} " + " | ".join(row), flush=True)
        del net, nets, x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
