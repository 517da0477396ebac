#!/usr/bin/env python3
"""Randomised differential test of DB post-processing on the GPU against the oracle: random maps made of rotated rectangles,
ellipses, thin strokes, blobs touching the frame, dips (holes) and background noise, random map / source sizes and random
thresholds; boxes must be identical integers, scores within 1e-6.
usage: python tools/fuzz_db.py [--cases 200] [--seed 0]"""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import pipeline_ref as P
from vse_amd import engine


def rand_map(rng, h, w):
    m = rng.uniform(0, rng.choice([0.05, 0.25, 0.31]), (h, w)).astype(np.float32)
    ys, xs = np.mgrid[0:h, 0:w]
    for _ in range(int(rng.integers(0, 14))):
        kind = rng.choice(["rect", "ellipse", "stroke", "edge"])
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        a, b = rng.uniform(3, w * 0.4), rng.uniform(1.5, h * 0.2)
        if kind == "stroke":
            b = rng.uniform(0.6, 2.5)
        if kind == "edge":
            cx, cy = rng.choice([0, w - 1]), rng.uniform(0, h)
        ang = rng.uniform(-np.pi / 2, np.pi / 2) if rng.random() < 0.6 else 0.0
        u = (xs - cx) * np.cos(ang) + (ys - cy) * np.sin(ang)
        v = -(xs - cx) * np.sin(ang) + (ys - cy) * np.cos(ang)
        inside = ((u / a) ** 2 + (v / b) ** 2 <= 1) if kind == "ellipse" else ((np.abs(u) <= a) & (np.abs(v) <= b))
        val = rng.uniform(0.31, 0.99) if rng.random() < 0.3 else rng.uniform(0.65, 0.99)
        if rng.random() < 0.3:                       # smooth fall-off instead of a flat top
            d = np.maximum(np.abs(u) / a, np.abs(v) / b)
            m = np.where(inside, np.maximum(m, (val * (1.15 - 0.5 * d)).astype(np.float32)), m)
        else:
            m[inside] = np.maximum(m[inside], np.float32(val))
        if rng.random() < 0.4:                       # dips
            for _ in range(int(rng.integers(1, 4))):
                dx, dy = int(cx + rng.uniform(-a, a)), int(cy + rng.uniform(-b, b))
                dw, dh = int(rng.integers(1, 6)), int(rng.integers(1, 5))
                m[max(dy, 0):max(dy + dh, 0), max(dx, 0):max(dx + dw, 0)] = rng.choice([0.0, 0.29, 0.3])
    return np.clip(m, 0, 1).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    ctx = engine.Context(0)
    rng = np.random.default_rng(a.seed)
    bad, nbox = [], 0
    done = 0
    while done < a.cases:
        h, w = int(rng.integers(8, 40)) * 4, int(rng.integers(8, 80)) * 4
        n = int(rng.integers(1, 5))
        maps = np.stack([rand_map(rng, h, w) for _ in range(n)])
        src_h, src_w = int(h * rng.uniform(0.5, 3.0)) + 1, int(w * rng.uniform(0.5, 3.0)) + 1
        kw = dict(thresh=float(rng.choice([0.3, 0.3, 0.2, 0.5])), box_thresh=float(rng.choice([0.6, 0.6, 0.4, 0.7])),
                  unclip_ratio=float(rng.choice([1.5, 1.5, 2.0, 1.2])))
        got = ctx.db_postprocess(torch.from_numpy(maps).cuda(), src_h, src_w, **kw)
        for f in range(n):
            rb, rs = P.db_postprocess(maps[f], src_h, src_w, **kw)
            gb, gs = got[f]
            nbox += len(rb)
            if gb.shape != rb.shape or not np.array_equal(gb, rb):
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"fuzz_db_fail_{a.seed}_{len(bad)}.npz"), prob=maps[f],
                                    src=np.array([src_h, src_w]), got=gb, **{k: np.float64(v) for k, v in kw.items()})
                bad.append(((h, w), (src_h, src_w), kw, f"boxes {len(gb)} vs oracle {len(rb)}",
                            [g.tolist() for g in gb if not any(np.array_equal(g, r) for r in rb)][:2],
                            [r.tolist() for r in rb if not any(np.array_equal(g, r) for g in gb)][:2]))
            elif len(rs) and np.abs(gs - rs).max() >= 1e-6:
                bad.append(((h, w), "score", float(np.abs(gs - rs).max())))
        done += n
    print(f"{done} maps, {nbox} boxes; failures: {len(bad)}")
    for b in bad[:30]:
        print("FAIL", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
