#!/usr/bin/env python3
"""A/B of the depthwise kernels (VSE_DW_TILE, csrc/simple_ops.hip): output hashes (the LDS-tile kernel must be bit-identical to the row
kernel) and net times.  Run once per setting, compare the lines:
    for t in 0 1 2; do VSE_DW_TILE=$t python tools/dw_tile_check.py; done"""
import hashlib
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vse_amd import engine, modelzoo

CASES = [("V4_ch_det_fast", (64, 544, 960), dict(hilo=True)), ("V4_ch_det_fast", (64, 544, 960), dict(hilo=True, chain=False)),
         ("V3_ch_det_fast", (64, 544, 960), dict(hilo=True)), ("V3_ch_det_fast", (3, 224, 352), dict(hilo=True, chain=False)),
         ("V4_ch_rec_fast", (56, 48, 896), dict(ragged=True)), ("V4_ch_rec", (56, 48, 896), dict(ragged=True)),
         ("V4_ch_det", (8, 544, 960), dict()), ("V3_en_rec_fast", (7, 48, 330), dict(ragged=True))]


def main():
    ctx = engine.Context(0)
    tag = os.environ.get("VSE_DW_TILE", "default")
    for mid, (n, h, w), kw in CASES:
        try:
            desc, wts = modelzoo.get_model(mid)
        except Exception as e:
            print(f"{mid}: skipped ({e})")
            continue
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
        outs = net.run(x, widths=widths) if widths is not None else net.run(x)
        torch.cuda.synchronize()
        hsh = hashlib.sha1(b"".join(o.cpu().numpy().tobytes() for o in outs)).hexdigest()[:16]
        if os.environ.get("VSE_DUMP"):
            np.save(os.path.join("gpurun_out", f"dump_{os.environ['VSE_DUMP']}_{mid}_{n}.npy"), outs[0].float().cpu().numpy())
        t0 = time.perf_counter()
        for _ in range(5):
            net.run(x, widths=widths) if widths is not None else net.run(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"VSE_DW_TILE={tag} {mid} {n}x{h}x{w} {kw}: {hsh}  {ms:.3f} ms")


if __name__ == "__main__":
    main()
