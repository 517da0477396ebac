#!/usr/bin/env python3
"""OP_CHAIN (csrc/chain.hip) on the GPU: the mobile detectors with and without chains against the fp32 oracle, on a few input
sizes (tile tails, single tiles), + timing at the bench shape.  usage: python tools/chain_check.py [--time]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import net_ref
from vse_amd import engine, ir

ctx = engine.Context(0)
bad = 0
for mid in ("V4_ch_det_fast", "V3_ch_det_fast"):
    if "--time-only" in sys.argv:
        break
    desc, w = net_ref.get_weights(mid)
    for (n, h, wd) in ((1, 96, 160), (2, 160, 256), (1, 224, 352), (3, 64, 64)):
        rng = np.random.default_rng(h)
        x = rng.uniform(-1, 1, (n, 3, h, wd)).astype(np.float16).astype(np.float32)
        ref = net_ref.run_graph(desc, w, x)[0].numpy()[:, 0]
        xt = torch.zeros((n, h, wd, 8), dtype=torch.float16, device="cuda")
        xt[..., :3] = torch.from_numpy(x.transpose(0, 2, 3, 1)).cuda().half()
        res = {}
        for chain in (False, True):
            net = engine.Net(ctx, desc, w, fetch_cols=(0,), hilo=True, chain=chain)
            out = net.run(xt)[0].float().cpu().numpy().reshape(n, h, wd)
            prog = net.program(n, h, wd)
            nchain = sum(int(o["kind"]) == ir.OP_CHAIN for o in prog.ops)
            res[chain] = (np.abs(out - ref).max(), len(prog.ops), nchain, out)
        d = np.abs(res[True][3] - res[False][3]).max()
        ok = res[True][0] < max(2e-2, 2 * res[False][0]) and np.isfinite(res[True][3]).all()
        bad += not ok
        print(f"{mid} {n}x{h}x{wd}: max|map - oracle| unchained {res[False][0]:.2e} ({res[False][1]} ops) chained {res[True][0]:.2e} "
              f"({res[True][1]} ops, {res[True][2]} chains); chained vs unchained {d:.2e} {'ok' if ok else 'FAIL'}", flush=True)
if "--time" in sys.argv or "--time-only" in sys.argv:
    for mid in ("V4_ch_det_fast", "V3_ch_det_fast"):
        desc, w = net_ref.get_weights(mid)
        x = (torch.rand((64, 544, 960, 8), device="cuda") * 2 - 1).half()
        x[..., 3:] = 0
        for chain in ((True,) if "--time-only" in sys.argv else (False, True)):
            net = engine.Net(ctx, desc, w, fetch_cols=(0,), hilo=True, chain=chain)
            for _ in range(2):
                net.run(x)
            ms, prog, names = net.profile(x)
            print(f"{mid} 64x544x960 chain={chain}: {ms.sum():.3f} ms, {len(prog.ops)} ops")
            if chain:
                for k, r in enumerate(prog.ops):
                    if int(r["kind"]) == ir.OP_CHAIN:
                        print(f"   op {k} {prog.names[k][:70]}: {ms[k]:.3f} ms, tiles {int(r['p'][0])}x{int(r['p'][1])}, LDS {int(r['p'][2])} B")
print("chain_check:", "FAILED" if bad else "all ok")
sys.exit(1 if bad else 0)
