import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from vse_amd import engine
ctx = engine.Context(0)
def run(cin, cout, k, h, w, n, reps):
    rng = np.random.default_rng(7)
    desc = {"model": "unit", "ops": [
        {"type": "feed", "in": {"X": ["feed"]}, "out": {"Out": ["x"]}, "attrs": {"col": 0}},
        {"type": "conv2d", "in": {"Input": ["x"], "Filter": ["w0"]}, "out": {"Output": ["t0"]}, "attrs": {"strides": [1, 1], "paddings": [0, 0], "groups": 1}},
        {"type": "conv2d", "in": {"Input": ["t0"], "Filter": ["w1"]}, "out": {"Output": ["t1"]}, "attrs": {"strides": [1, 1], "paddings": [k[0] // 2, k[1] // 2], "groups": 1}},
        {"type": "relu", "in": {"X": ["t1"]}, "out": {"Out": ["t2"]}, "attrs": {}},
        {"type": "fetch", "in": {"X": ["t2"]}, "out": {"Out": ["fetch"]}, "attrs": {"col": 0}}],
        "params": {"w0": {"dims": [cin, 3, 1, 1], "dtype": 5}, "w1": {"dims": [cout, cin, k[0], k[1]], "dtype": 5}},
        "var_shapes": {"t0": [-1, cin, -1, -1], "t1": [-1, cout, -1, -1]}}
    wts = {"w0": rng.standard_normal((cin, 3, 1, 1)).astype(np.float32), "w1": (rng.standard_normal((cout, cin, k[0], k[1])) / np.sqrt(cin * k[0] * k[1])).astype(np.float32)}
    net = engine.Net(ctx, desc, wts)
    x = (torch.rand((n, h, w, 8), device="cuda") * 2 - 1).half(); x[..., 3:] = 0
    first = net.run(x)[0].clone()
    side = torch.cuda.Stream(); junk = torch.rand((4096, 4096), device="cuda"); bad = 0
    for rep in range(reps):
        with torch.cuda.stream(side):
            for _ in range(rep % 4): junk = junk @ junk * 1e-4
        if not torch.equal(net.run(x)[0], first): bad += 1
    torch.cuda.synchronize()
    print(cin, cout, k, h, w, n, "reps", reps, "mismatches", bad, flush=True)
run(256, 64, (9, 9), 136, 240, 16, 150)
run(64, 64, (9, 9), 68, 120, 64, 150)
run(128, 128, (3, 3), 136, 240, 16, 200)
run(64, 64, (3, 3), 272, 480, 8, 150)
run(160, 160, (3, 3), 68, 120, 32, 200)
run(192, 192, (3, 3), 6, 192, 28, 300)
run(32, 32, (5, 5), 136, 240, 16, 150)
run(96, 24, (3, 3), 136, 240, 32, 200)      # conv_c3n32_kernel
run(32, 32, (3, 3), 136, 240, 32, 200)


def run_rec(mid, n, widths_lo, wt, reps):
    """a ragged recogniser batch beside unrelated work on another stream: identical bits every time"""
    sys.path.insert(0, os.getcwd())
    from vse_amd import modelzoo
    desc, wts = modelzoo.get_model(mid, seed=1)
    net = engine.Net(ctx, desc, wts, want_probs=False, ragged=True)
    h = 32 if mid.startswith("V2") else 48
    rng = np.random.default_rng(3)
    widths = rng.integers(widths_lo, wt + 1, n).astype(np.int32)
    x = (torch.rand((n, h, wt, 8), device="cuda") * 2 - 1).half(); x[..., 3:] = 0
    for i, wi in enumerate(widths):
        x[i, :, int(wi):] = 0
    first = net.run(x, widths=widths)[-1].clone()
    side = torch.cuda.Stream(); junk = torch.rand((4096, 4096), device="cuda"); bad = 0
    for rep in range(reps):
        with torch.cuda.stream(side):
            for _ in range(rep % 4): junk = junk @ junk * 1e-4
        if not torch.equal(net.run(x, widths=widths)[-1], first): bad += 1
    torch.cuda.synchronize()
    print(mid, n, wt, "reps", reps, "mismatches", bad, flush=True)


run_rec("V4_ch_rec", 24, 400, 768, 60)
run_rec("V4_en_rec_fast", 48, 330, 640, 100)
run_rec("V2_ch_rec", 40, 330, 640, 60)
