#!/usr/bin/env python3
"""Per-op timing of one model on the GPU (HIP events inside vse_plan_profile).
usage: python tools/gpu_profile_net.py MODEL N H W [--top K] [--hilo [--no-chain]] [--ragged [--wmin W0]]
--ragged: a recogniser plan for ragged batches; sample widths are spread evenly over [W0 (default 320), W]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vse_amd import engine, ir, modelzoo


def op_macs(r):
    if int(r["kind"]) != ir.OP_CONV:
        return 0.0
    p = r["p"]
    o = r["out"]
    m = int(o["n"]) * int(o["h"]) * int(o["w"])
    if int(r["flags"]) & ir.F_PIXSHUF:
        m //= 4
    return m * float(p[ir.P_COUT]) * float(p[ir.P_KTOT])   # padded (executed) MACs


def view_bytes(v):
    return float(v["n"]) * float(v["h"]) * float(v["w"]) * float(v["c"]) * float(v["esize"]) if int(v["n"]) > 0 else 0.0


def op_sol_ms(r):
    """Speed-of-light time of one op: max(executed MACs / 2.5 PFLOP/s dense fp16, bytes of its views once / 8 TB/s)."""
    byts = sum(view_bytes(r[k]) for k in ("in0", "in1", "in2", "out", "out2"))
    if int(r["kind"]) == ir.OP_CONV:
        byts += 2.0 * float(r["p"][ir.P_COUT]) * float(r["p"][ir.P_KTOT])
    return 1e3 * max(2 * op_macs(r) / 2.5e15, byts / 8e12), byts


def main():
    mid = sys.argv[1]
    n, h, w = (int(v) for v in sys.argv[2:5])
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    desc, wts = modelzoo.get_model(mid)
    ctx = engine.Context(0)
    ragged = "--ragged" in sys.argv
    net = engine.Net(ctx, desc, wts, want_probs=False, hilo="--hilo" in sys.argv, ragged=ragged,
                     chain=False if "--no-chain" in sys.argv else None)       # --no-chain: hi + lo weights, layer by layer, no pair tensors
    x = (torch.rand((n, h, w, 8), device="cuda") * 2 - 1).half()
    x[..., 3:] = 0
    widths = None
    if ragged:
        w0 = int(sys.argv[sys.argv.index("--wmin") + 1]) if "--wmin" in sys.argv else 320
        widths = np.linspace(w0, w, n).astype(np.int32)
        for i, wi in enumerate(widths):
            x[i, :, int(wi):] = 0
        print("sample widths:", widths.tolist())
    for _ in range(2):
        net.run(x, widths=widths)
    torch.cuda.synchronize()
    if "--after-det" in sys.argv:
        # the recogniser as the pipeline meets it: right behind a 64-frame server-detector pass (power / clock state, caches)
        ddesc, dw = modelzoo.get_model("V4_ch_det")
        dnet = engine.Net(ctx, ddesc, dw, fetch_cols=(0,))
        dx = (torch.rand((64, 544, 960, 8), device="cuda") * 2 - 1).half()
        dnet.run(dx)
        torch.cuda.synchronize()
        samples = []
        for _ in range(3):
            dnet.run(dx)
            samples.append(net.profile(x, widths=widths)[0])
        ms, prog, names = net.profile(x, widths=widths)
        print("totals right behind a detector pass (ms):", [round(float(v.sum()), 3) for v in samples])
        ms = np.minimum.reduce(samples)
    else:
        ms, prog, names = net.profile(x, widths=widths)
        ms2, _, _ = net.profile(x, widths=widths)
        ms = np.minimum(ms, ms2)
    tot = ms.sum()
    print(f"{mid} N={n} {h}x{w}: {len(ms)} ops, total {tot:.3f} ms, algorithmic {prog.gmacs:.2f} GMAC -> "
          f"{2 * prog.gmacs / tot:.1f} TFLOP/s effective, ws {prog.ws_bytes / 1e9:.2f} GB")
    bykind = {}
    for k, r in enumerate(prog.ops):
        bykind.setdefault(int(r["kind"]), [0.0, 0])
        bykind[int(r["kind"])][0] += ms[k]
        bykind[int(r["kind"])][1] += 1
    print("by kind:", {k: (round(v[0], 3), v[1]) for k, v in sorted(bykind.items())})
    sol = np.array([op_sol_ms(r)[0] for r in prog.ops])
    print(f"speed of light (per op max(MFMA, HBM), summed): {sol.sum():.3f} ms = {100 * sol.sum() / tot:.1f} % of the measured total; "
          f"MFMA-bound ops {sol[[2 * op_macs(r) / 2.5e15 * 1e3 >= op_sol_ms(r)[0] for r in prog.ops]].sum():.3f} ms")
    order = np.argsort(-(ms - sol) if "--by-gap" in sys.argv else -ms)[:top]
    for k in order:
        r = prog.ops[k]
        p = r["p"]
        o = r["out"]
        macs = op_macs(r)
        extra = ""
        if int(r["kind"]) == ir.OP_CONV:
            extra = (f"k{p[0]}x{p[1]} s{p[2]} cin{p[ir.P_CINP]} N{p[ir.P_COUT]} K{p[ir.P_KTOT]} "
                     f"{2 * macs / ms[k] / 1e9:.0f} TF/s(padded)")
        print(f"  op{k:3d} kind={int(r['kind']):2d} {ms[k]:8.3f} ms {100 * ms[k] / tot:5.1f}% sol {sol[k]:6.3f} ({100 * sol[k] / max(ms[k], 1e-9):3.0f}%) out[{o['n']},{o['h']},{o['w']},{o['c']}] "
              f"{prog.names[k][:28]:28s} {names[k]:44s} {extra}")


if __name__ == "__main__":
    main()
