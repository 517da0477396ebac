#!/usr/bin/env python3
"""Randomised differential test of the pre-processing kernels on the GPU, bit-exact against the oracle:
  det pre-process   random frame sizes (tiny, odd, non-multiples of 32, 4K) of noise frames;
  rec pre-process   random quads on noise frames: rotated / skewed / partly outside the frame (replicate border) / 1-3 pixel
                    sides / tall boxes (rotated by 90 degrees) / very wide boxes, random canvas widths.
usage: python tools/fuzz_prepost.py [--cases 200] [--seed 0]"""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import pipeline_ref as P
from vse_amd import engine, pipeline


def rand_quad(rng, h, w):
    kind = rng.choice(["box", "rot", "skew", "tiny", "tall", "wide", "outside"])
    cx, cy = rng.uniform(0, w), rng.uniform(0, h)
    bw, bh = rng.uniform(8, w * 0.6), rng.uniform(4, h * 0.3)
    if kind == "tiny":
        bw, bh = rng.uniform(1, 4), rng.uniform(1, 4)
    elif kind == "tall":
        bw, bh = rng.uniform(4, 20), rng.uniform(30, max(31.0, h * 0.8))
    elif kind == "wide":
        bw, bh = rng.uniform(w * 0.5, w * 1.1), rng.uniform(4, 16)
    ang = rng.uniform(-0.6, 0.6) if kind in ("rot", "skew", "outside") else rng.uniform(-0.03, 0.03)
    c, s = np.cos(ang), np.sin(ang)
    q = np.array([[-bw / 2, -bh / 2], [bw / 2, -bh / 2], [bw / 2, bh / 2], [-bw / 2, bh / 2]])
    if kind == "skew":
        q += rng.uniform(-0.2, 0.2, (4, 2)) * [bw, bh]
    q = q @ np.array([[c, s], [-s, c]]) + [cx, cy]
    if kind != "outside":
        q[:, 0] = np.clip(q[:, 0], 0, w - 1)
        q[:, 1] = np.clip(q[:, 1], 0, h - 1)
    if rng.random() < 0.7:
        q = np.rint(q)                              # DB boxes are integer-valued
    return kind, q.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    ctx = engine.Context(0)
    rng = np.random.default_rng(a.seed)
    bad = []
    # ---- det pre-process -------------------------------------------------------------------------------------------
    sizes = [(1, 1), (7, 5), (31, 33), (32, 32), (33, 31), (100, 37), (37, 1000), (480, 854), (1081, 1921), (2160, 3840)]
    sizes += [(int(rng.integers(1, 1300)), int(rng.integers(1, 2200))) for _ in range(max(4, a.cases // 10))]
    for hw in sizes:
        frames = rng.integers(0, 256, (1,) + hw + (3,), dtype=np.uint8)
        rh, rw = P.det_resize_shape(*hw)
        try:
            got = ctx.det_preprocess(torch.from_numpy(frames).cuda(), rh, rw).cpu().numpy()
            ref = P.det_preprocess(frames[0])[0][0].transpose(1, 2, 0).astype(np.float16)
            if got.shape[1:3] != ref.shape[:2] or not np.array_equal(got[0, ..., :3], ref):
                bad.append(("det", hw, "mismatch"))
        except Exception as e:                                        # noqa: BLE001
            bad.append(("det", hw, type(e).__name__ + ": " + str(e)[:100]))
    # ---- rec pre-process -------------------------------------------------------------------------------------------
    kinds = {}
    done = 0
    while done < a.cases:
        h, w = int(rng.integers(40, 400)), int(rng.integers(60, 700))
        frames = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
        dev = torch.from_numpy(frames).cuda()
        img_w = int(rng.choice([320, 384, 512, 640, 1000]))
        quads, specs = [], []
        for _ in range(int(rng.integers(1, 9))):
            kind, q = rand_quad(rng, h, w)
            f = int(rng.integers(0, 2))
            cw, ch, rot = pipeline.crop_geometry(q)
            if cw < 1 or ch < 1:
                continue
            iw, ih = (ch, cw) if rot else (cw, ch)
            rw = P.rec_resized_width(iw, ih, img_w)
            quads.append((kind, f, q))
            specs.append(dict(quad=q, frame=f, crop_w=cw, crop_h=ch, resized_w=rw, rotate=rot))
        if not specs:
            continue
        done += len(specs)
        try:
            got = ctx.rec_preprocess(dev, specs, 48, img_w).cpu().numpy()
        except Exception as e:                                        # noqa: BLE001
            bad.append(("rec", [q.tolist() for _, _, q in quads], type(e).__name__ + ": " + str(e)[:100]))
            continue
        for k, (kind, f, q) in enumerate(quads):
            kinds[kind] = kinds.get(kind, 0) + 1
            ref = P.resize_norm_img(P.get_rotate_crop_image(frames[f], q), img_w).transpose(1, 2, 0).astype(np.float16)
            if not np.array_equal(got[k, ..., :3], ref) or not np.all(got[k, :, specs[k]["resized_w"]:, :] == 0):
                d = np.abs(got[k, ..., :3].astype(np.float32) - ref.astype(np.float32))
                bad.append(("rec", kind, (h, w, img_w), q.tolist(), f"max diff {d.max():.4f} at {int((d > 0).sum())} values"))
    print(f"det sizes {len(sizes)}, rec crops {done} {kinds}; failures: {len(bad)}")
    for b in bad[:40]:
        print("FAIL", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
