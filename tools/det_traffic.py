#!/usr/bin/env python3
"""HBM bytes per frame of ONE detector program from the FETCH_SIZE / WRITE_SIZE counters (counter bytes of the
mobile detector against SURVEY §8(d)'s 141.7 MB per frame of unfused layer-wise fp16 activation traffic).

  run (on the GPU box, once per counter, separate passes as MI355X_MICROARCH.md prescribes):
      rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR_F -o t -- python tools/det_traffic.py run MODEL [--layerwise] [--plain]
      rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d DIR_W -o t -- python tools/det_traffic.py run MODEL [--layerwise] [--plain]
  sum (anywhere):
      python tools/det_traffic.py sum DIR_F DIR_W

`run` executes RUNS forward passes of 64 x 544 x 960 and nothing else on the GPU (the input is filled by one host copy), so the sum of
the counters over every launch of the process, divided by RUNS x 64, is the detector's traffic per frame.
bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (KiB units; gfx950 FETCH_SIZE half-count correction)."""
import csv
import glob
import json
import os
import sys

RUNS, N, H, W = 4, 64, 544, 960


def run(argv):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    import torch
    from vse_amd import engine, modelzoo
    mid = argv[0]
    desc, wts = modelzoo.get_model(mid)
    ctx = engine.Context(0)
    plain = "--plain" in argv                      # single fp16 weights: no pair tensors, no chains (the round-3 program)
    net = engine.Net(ctx, desc, wts, fetch_cols=(0,), hilo=not plain, chain=False if "--layerwise" in argv else None)
    x = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, (N, H, W, 8)).astype(np.float16))
    x[..., 3:] = 0
    x = x.cuda()
    for _ in range(RUNS):
        net.run(x)
    torch.cuda.synchronize()
    print(f"{mid}: {RUNS} passes of {N}x{H}x{W}, {len(net.program(N, H, W).ops)} ops")


def total(d):
    s, n, per = 0.0, 0, {}
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            v = float(r["Counter_Value"])
            s += v
            n += 1
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            per[k] = per.get(k, 0.0) + v
    return s, n, per


def main():
    if sys.argv[1] == "run":
        return run(sys.argv[2:])
    f, nf, pf = total(sys.argv[2])
    w, nw, pw = total(sys.argv[3])
    frames = RUNS * N
    by = {k: round((2 * pf.get(k, 0.0) + pw.get(k, 0.0)) * 1024 / frames / 1e6, 3) for k in set(pf) | set(pw)}
    print(json.dumps({"frames": frames, "launches": [nf, nw],
                      "fetch_MB_per_frame": round(2 * f * 1024 / frames / 1e6, 2), "write_MB_per_frame": round(w * 1024 / frames / 1e6, 2),
                      "hbm_MB_per_frame": round((2 * f + w) * 1024 / frames / 1e6, 2),
                      "by_kernel_MB_per_frame": dict(sorted(by.items(), key=lambda kv: -kv[1]))}, indent=1))


if __name__ == "__main__":
    main()
