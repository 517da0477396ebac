#!/usr/bin/env python3
"""Outputs of whole networks under two settings of a library environment switch (read once per process, hence child processes):
identical bits?  and the per-net GPU time of each.   usage: python tools/ab_env_digest.py VAR [value_a value_b]"""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = (("V3_ch_det_fast", (8, 3, 544, 960), True), ("V4_ch_det_fast", (8, 3, 544, 960), True), ("V4_ch_det", (4, 3, 288, 512), False),
         ("V4_ch_rec", (9, 3, 48, 480), False), ("V4_en_rec_fast", (16, 3, 48, 352), False), ("V2_ch_rec", (6, 3, 32, 256), False))


def child():
    import hashlib
    import numpy as np
    import torch
    from vse_amd import engine, modelzoo
    from oracle import ir_emul
    ctx = engine.Context(0)
    for mid, shape, hilo in CASES:
        desc, w = modelzoo.get_model(mid)
        net = engine.Net(ctx, desc, w, want_probs=False, hilo=hilo)
        x = np.random.default_rng(11).uniform(-1, 1, shape).astype(np.float32)
        xt = torch.from_numpy(ir_emul.to_nhwc8(x).astype(np.float16)).cuda()
        h = hashlib.sha256()
        for o in net.run(xt):
            h.update(np.ascontiguousarray(o.cpu().numpy()).tobytes())
        ms = min(float(net.profile(xt)[0].sum()) for _ in range(3))
        print("DIGEST", mid, h.hexdigest()[:20], f"{ms:.3f} ms", flush=True)


def main():
    var = sys.argv[1]
    vals = sys.argv[2:4] if len(sys.argv) >= 4 else ["0", "1"]
    out = {}
    for v in vals:
        r = subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, **{var: v}), capture_output=True, text=True)
        out[v] = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("DIGEST")]
        if r.returncode != 0:
            print(r.stderr[-1500:])
    same = True
    for a, b in zip(out[vals[0]], out[vals[1]]):
        ok = a[2] == b[2]
        same &= ok
        print(f"{a[1]:18s} {var}={vals[0]}: {a[3]:>8s} ms   {var}={vals[1]}: {b[3]:>8s} ms   {'identical' if ok else 'DIFFERENT'}")
    sys.exit(0 if same and len(out[vals[0]]) == len(CASES) else 1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        main()
