#!/usr/bin/env python3
"""What the engine's recogniser outputs achieve against the fp32 oracle, in the units the parity tests assert: per model,
max / median |delta log p| over ALL classes and over the oracle's top-5, max |delta p|, relative error of the per-step max probability, the
number of arg-max flips and the largest oracle top-2 log-margin at which one happens.  GPU box:  python tools/rec_margin_study.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import ir_emul, net_ref
from vse_amd import engine

CASES = [("V4_ch_rec", (3, 3, 48, 200)), ("V4_ch_rec", (2, 3, 48, 896)), ("V4_ch_rec_fast", (2, 3, 48, 320)), ("V4_en_rec_fast", (6, 3, 48, 352)),
         ("V3_ch_rec_fast", (2, 3, 48, 160)), ("V3_latin_rec_fast", (2, 3, 48, 160)), ("V2_ch_rec", (2, 3, 32, 128))]


def main():
    ctx = engine.Context(0)
    for mid, shape in CASES:
        desc, w = net_ref.get_weights(mid)
        x = np.random.default_rng(0).uniform(-1, 1, shape).astype(np.float16).astype(np.float32)
        ref = net_ref.run_graph(desc, w, x)[0].numpy().astype(np.float64)
        net = engine.Net(ctx, desc, w, want_probs=True)
        outs = [o.cpu().numpy() for o in net.run(torch.from_numpy(ir_emul.to_nhwc8(x).astype(np.float16)).cuda())]
        probs = outs[0][:, 0].astype(np.float64)
        idx = outs[-1].view(np.int32)[:, 0, :, 0]
        tiny = 1e-30
        dl = np.abs(np.log(np.maximum(probs, tiny)) - np.log(np.maximum(ref, tiny)))
        live = ref > 1e-12
        order = np.argsort(-ref, -1)[..., :5]
        dl5 = np.take_along_axis(dl, order, -1)
        srt = np.sort(ref, -1)
        gap = np.log(srt[..., -1]) - np.log(srt[..., -2])
        flips = idx != ref.argmax(-1)
        maxp_rel = np.abs(probs.max(-1) - ref.max(-1)) / ref.max(-1)
        print(f"{mid:18s} {str(shape):18s} classes {ref.shape[-1]:5d}  median max-p {np.median(ref.max(-1)):.4f}  "
              f"|dlogp| all: max {dl[live].max():.2e} median {np.median(dl[live]):.2e}  top-5: max {dl5.max():.2e}  "
              f"|dp| max {np.abs(probs - ref).max():.2e}  max-p rel err max {maxp_rel.max():.2e}  "
              f"flips {int(flips.sum())}/{flips.size} (largest gap at a flip {gap[flips].max() if flips.any() else 0:.2e}); "
              f"steps with gap < 2e-2: {float((gap < 2e-2).mean()):.3f}, < 5e-2: {float((gap < 5e-2).mean()):.3f}", flush=True)


if __name__ == "__main__":
    main()
