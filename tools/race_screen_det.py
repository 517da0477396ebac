#!/usr/bin/env python3
"""Race screen of the whole server detector (the persistent resident-weight head kernel included): the same frames many times,
beside unrelated matrix work on another stream and with different batch sizes (= different persistent tile schedules); the
probability maps must be identical bit for bit every time, and identical between the head kernel's two forms.
usage: python tools/race_screen_det.py [reps]"""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import hashlib
    import numpy as np
    import torch
    from vse_amd import engine, modelzoo
    reps = int(sys.argv[2])
    torch.manual_seed(0)
    ctx = engine.Context(0)
    desc, w = modelzoo.get_model("V4_ch_det")
    net = engine.Net(ctx, desc, w, fetch_cols=(0,))
    side = torch.cuda.Stream()
    junk = torch.rand((4096, 4096), device="cuda")
    for n, h, wd in ((64, 544, 960), (5, 544, 960), (3, 288, 512), (2, 160, 224), (1, 96, 160)):
        x = (torch.rand((n, h, wd, 8), device="cuda") * 2 - 1).half()
        x[..., 3:] = 0
        first = net.run(x)[0].clone()
        bad = 0
        for rep in range(reps if n < 64 else max(4, reps // 8)):
            with torch.cuda.stream(side):
                for _ in range(rep % 4):
                    junk = junk @ junk * 1e-4
            if not torch.equal(net.run(x)[0], first):
                bad += 1
        torch.cuda.synchronize()
        print("RESULT", n, h, wd, "mismatches", bad, hashlib.sha256(first.cpu().numpy().tobytes()).hexdigest()[:16], flush=True)


def main():
    reps = sys.argv[1] if len(sys.argv) > 1 else "60"
    out = {}
    for form in ("1", "0"):
        r = subprocess.run([sys.executable, __file__, "--child", reps], env=dict(os.environ, VSE_HEAD_RESIDENT=form), capture_output=True, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
        print(f"VSE_HEAD_RESIDENT={form}:")
        for ln in lines:
            print("  ", ln)
        if r.returncode != 0:
            print(r.stderr[-1500:])
        out[form] = [ln.split()[-1] for ln in lines], all(ln.split()[5] == "0" for ln in lines) and len(lines) == 5
    ok = out["1"][1] and out["0"][1] and out["1"][0] == out["0"][0]
    print("race screen:", "clean, both forms identical" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        main()
