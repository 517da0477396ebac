#!/usr/bin/env python3
"""Box / string parity of the HIP pipeline against the oracle over MANY synthetic frames (the tests pin a few):
usage: python tools/parity_sweep.py [n_frames] [height] [width]   -> counts of identical boxes, min IoU, string matches."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import net_ref
from oracle import pipeline_ref as P
from vse_amd import engine, pipeline, synth


def iou(a, b):
    ax0, ay0, ax1, ay1 = a[:, 0].min(), a[:, 1].min(), a[:, 0].max(), a[:, 1].max()
    bx0, by0, bx1, by1 = b[:, 0].min(), b[:, 1].min(), b[:, 0].max(), b[:, 1].max()
    iw, ih = max(0, min(ax1, bx1) - max(ax0, bx0)), max(0, min(ay1, by1) - max(ay0, by0))
    u = (ax1 - ax0) * (ay1 - ay0) + (bx1 - bx0) * (by1 - by0) - iw * ih
    return iw * ih / u if u > 0 else 1.0


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    sys.argv = [sys.argv[0]] + args + [a for a in sys.argv[1:] if a.startswith("-")]
    n = int(sys.argv[1]) if len(args) > 0 else 32
    h = int(sys.argv[2]) if len(args) > 1 else 1080
    w = int(sys.argv[3]) if len(args) > 2 else 1920
    ctx = engine.Context(0)
    det = net_ref.get_weights("V3_ch_det_fast")
    rec = net_ref.get_weights("V4_en_rec_fast")
    dw = "fp16" if "--fp16" in sys.argv else "auto"          # auto = fp16 hi + lo weight pairs for this (mobile) detector
    pipe = pipeline.OcrPipeline(ctx, det, rec, P.en_charset(), rec_mode="reference", det_weights=dw)
    print("detector weights:", pipe.det_weights)
    frames = synth.make_frames(n, h, w, seed=777, p_two_lines=0.5)
    got = pipe.ocr(torch.from_numpy(frames).cuda())
    nb = same = cnt_mismatch = 0
    ious = []
    for f in range(n):
        x, _ = P.det_preprocess(frames[f])
        prob = net_ref.run_graph(det[0], det[1], x)[0].numpy()[0, 0]
        rb = P.sorted_boxes(P.db_postprocess(prob, h, w)[0])
        gb = got[f][0]
        if len(gb) != len(rb):
            cnt_mismatch += 1
            continue
        for a, b in zip(gb, rb):
            a, b = np.asarray(a), np.asarray(b)
            nb += 1
            same += int(np.array_equal(a, b))
            ious.append(iou(a, b))
            if ious[-1] < 0.99 or "-v" in sys.argv:
                if ious[-1] < 0.999:
                    maps = pipe.det_maps(torch.from_numpy(frames[f:f + 1]).cuda())[0].cpu().numpy()
                    d = np.abs(maps - prob)
                    print(f"frame {f}: hip {a.astype(int).tolist()} oracle {b.astype(int).tolist()} IoU {ious[-1]:.4f}; prob map max |diff| {d.max():.4f}, "
                          f"pixels on opposite sides of 0.3: {int(((maps > 0.3) != (prob > 0.3)).sum())}")
    ious = np.asarray(ious)
    print(f"{n} frames {h}x{w}: {nb} boxes, {same} identical, {int((ious < 0.99).sum())} with IoU < 0.99 (min {ious.min():.4f}), "
          f"{cnt_mismatch} frames with a different box count")


if __name__ == "__main__":
    main()
