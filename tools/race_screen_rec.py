#!/usr/bin/env python3
"""Run-to-run determinism of whole nets under a disturbed GPU (a side stream of GEMMs of varying length): every repetition must return
the bits of the first one.  usage: python tools/race_screen_rec.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vse_amd import engine, modelzoo

ctx = engine.Context(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for mid, (n, h, w), kw in [("V4_ch_rec", (56, 48, 896), dict(ragged=True)), ("V4_ch_rec_fast", (56, 48, 896), dict(ragged=True)),
                           ("V4_ch_det_fast", (16, 544, 960), dict(hilo=True)), ("V4_ch_det", (8, 544, 960), dict())]:
    desc, wts = modelzoo.get_model(mid)
    det = "det" in mid
    net = engine.Net(ctx, desc, wts, want_probs=not det, **({"fetch_cols": (0,)} if det else {}), **kw)
    x = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, (n, h, w, 8)).astype(np.float16))
    x[..., 3:] = 0
    x = x.cuda()
    widths = None
    if kw.get("ragged"):
        widths = np.linspace(max(32, w // 3), w, n).astype(np.int32)
        for i, wi in enumerate(widths):
            x[i, :, int(wi):] = 0
    run = (lambda: net.run(x, widths=widths)) if widths is not None else (lambda: net.run(x))
    first = [o.clone() for o in run()]
    side = torch.cuda.Stream()
    junk = torch.rand((2048, 2048), device="cuda")
    bad = 0
    for rep in range(reps):
        with torch.cuda.stream(side):
            for _ in range(rep % 4):
                junk = junk @ junk * 1e-4
        outs = run()
        if not all(torch.equal(a, b) for a, b in zip(outs, first)):
            bad += 1
    torch.cuda.synchronize()
    print(f"{mid} {n}x{h}x{w} {kw}: {reps} repetitions, {bad} mismatches", flush=True)
