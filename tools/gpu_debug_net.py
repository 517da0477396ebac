#!/usr/bin/env python3
"""GPU bring-up aid: run a compiled program on the device with buffer reuse disabled and compare EVERY op's
output view against the CPU emulator of the same program (oracle/ir_emul.py, fp16-rounded mode).
usage: python tools/gpu_debug_net.py MODEL [N H W]"""
import ctypes as C
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import net_ref, ir_emul
from vse_amd import compiler, engine, ir


def main():
    mid = sys.argv[1]
    args = [a for a in sys.argv[2:] if not a.startswith("-")]
    shape = tuple(int(v) for v in args[:3]) if len(args) >= 3 else \
        ((1, 64, 96) if "_det" in mid else ((2, 32, 160) if mid.startswith("V2") else (2, 48, 160)))
    n, h, w = shape
    desc, wts = net_ref.get_weights(mid)
    x = np.random.default_rng(0).uniform(-1, 1, (n, 3, h, w)).astype(np.float32)
    x8 = ir_emul.to_nhwc8(x)
    hilo = "--hilo" in sys.argv          # fp16 hi + lo weights; 1x1 / depthwise chains (OP_CHAIN) with them
    prog = compiler.compile_model(desc, wts, n, h, w, reuse=False, hilo=hilo)
    em = ir_emul.Emulator(prog, round_f16=True)
    ref_outs = em.run(x8)
    ctx = engine.Context(0)
    net = engine.Net(ctx, desc, wts, hilo=hilo)
    net.plans[(n, h, w)] = [prog, None]
    net.store = prog.weights
    xt = torch.from_numpy(x8.astype(np.float16)).cuda()
    outs = net.run(xt)
    torch.cuda.synchronize()
    ws = net.ws[((n, h, w), 0)][0].cpu().numpy()
    bad = 0
    views = [(k, r, r[slot]) for k, r in enumerate(prog.ops)
             for slot in (("out", "out2", "in2") if int(r["kind"]) == ir.OP_CHAIN else ("out",)) if int(r[slot]["n"]) > 0]
    for k, r, v in views:
        if int(v["arena"]) != ir.ARENA_WS:
            continue
        nn, hh, ww, cc, ld, es = (int(v[q]) for q in ("n", "h", "w", "c", "ld", "esize"))
        dt = np.float16 if es == 2 else np.float32
        base = int(v["off"])
        cnt = (nn * hh * ww - 1) * ld + cc
        idx = (np.arange(nn * hh * ww)[:, None] * ld + np.arange(cc)[None, :])
        got = ws[base:base + cnt * es].view(dt)[idx].astype(np.float32)
        exp = em.ws[base:base + cnt * es].view(dt)[idx].astype(np.float32)
        err = np.abs(got - exp)
        tol = 2e-2 + 1e-2 * np.abs(exp)
        nbad = int((err > tol).sum()) + int(np.isnan(got).sum())
        status = "ok " if nbad == 0 else "BAD"
        if nbad or "-v" in sys.argv:
            print(f"{status} op{k:3d} kind={int(r['kind'])} {prog.names[k][:40]:40s} out[{nn},{hh},{ww},{cc}] "
                  f"maxerr={np.nanmax(err):.4g} maxref={np.abs(exp).max():.3g} nbad={nbad}")
        bad += nbad > 0
    for o, ro, meta in zip(outs, ref_outs, prog.outputs):
        g = o.cpu().numpy()
        if meta["kind"] == "idx_maxp":
            gi, ri = g.view(np.int32)[..., 0], ro.view(np.int32)[..., 0]
            print("output idx match", (gi == ri).mean(), "maxp err", np.abs(g[..., 1] - ro[..., 1]).max())
        else:
            print("output", meta["kind"], "maxerr", np.abs(g - ro).max(), "ref range", ro.min(), ro.max())
    print(f"{mid}: {len(prog.ops)} ops, {bad} mismatching")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
