#!/usr/bin/env python3
"""Randomised differential test of the conv kernels on the GPU: single-conv graphs with random channels / filter / stride /
map size / batch (drawn around the tile, chunk and cout-tile boundaries of every kernel family) through the engine, against the
torch-CPU fp32 oracle — the body of tests/test_gpu_nets.py::test_conv_shapes over a few hundred seeds instead of its fixed list.
usage: python tools/fuzz_conv.py [--cases 300] [--seed 0]     (exit code 1 and the failing tuples on any mismatch)"""
import argparse
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def draw(rng):
    fam = rng.choice(["c3", "col", "pw", "gemm1x1", "gemm", "stem", "any"], p=[0.25, 0.2, 0.1, 0.12, 0.13, 0.05, 0.15])
    pick = lambda xs: int(xs[rng.integers(len(xs))])
    if fam == "c3":
        cin = pick([16, 32, 48, 64, 96, 128, 160, 192, 224, 256, 320])
        cout = pick([8, 16, 24, 32, 40, 64, 72, 96, 128, 160, 192])
        k, s, p = (3, 3), (1, 1), (1, 1)
        h, w = pick([1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15, 16, 17, 24, 31, 33, 40]), pick([1, 7, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 200, 257])
    elif fam == "col":
        kh, kw = pick([5, 7, 9]), pick([5, 7, 9])
        cin, cout = pick([16, 32, 48, 64, 128, 256]), pick([8, 16, 24, 32, 40, 64])
        k, s, p = (kh, kw), (1, 1), (kh // 2, kw // 2)
        h, w = pick([4, 8, 9, 15, 16, 17, 24, 31, 32, 33, 40]), pick([5, 31, 32, 33, 47, 48, 49, 64, 95, 96, 97])
    elif fam == "pw":
        cin, cout = pick([8, 16, 24, 32, 48, 64]), pick([8, 16, 24, 32, 40, 48, 64])
        k, s, p = (1, 1), (1, 1), (0, 0)
        h, w = pick([1, 3, 9, 16, 17, 33]), pick([1, 5, 16, 31, 40, 50, 64])
    elif fam == "gemm1x1":
        cin, cout = pick([96, 128, 160, 256, 320, 512, 896]), pick([32, 64, 72, 128, 200, 256, 320, 512])
        k, s, p = (1, 1), tuple([pick([1, 1, 1, 2])] * 2), (0, 0)
        h, w = pick([1, 3, 8, 13, 15, 17, 34]), pick([7, 17, 23, 30, 40, 60])
    elif fam == "gemm":
        kh, kw = pick([1, 2, 3, 5]), pick([1, 2, 3, 5])
        cin, cout = pick([32, 64, 96, 128, 160]), pick([16, 32, 48, 64, 72, 128, 136, 200])
        k, s, p = (kh, kw), (pick([1, 2]), pick([1, 2])), (pick([0, kh // 2]), pick([0, kw // 2]))
        h, w = pick([6, 10, 13, 18, 21]), pick([6, 12, 17, 22, 37])
    elif fam == "stem":
        cin, cout = pick([3, 4]), pick([16, 24, 32, 40, 48, 64])
        k, s, p = (3, 3), tuple([pick([1, 2])] * 2), (1, 1)
        h, w = pick([8, 16, 19, 37, 48]), pick([8, 32, 45, 64, 70])
    else:
        kh, kw = pick([1, 1, 3, 3, 5, 7]), pick([1, 1, 3, 3, 5, 7])
        cin, cout = pick([3, 8, 16, 24, 40, 72, 100, 128]), pick([8, 16, 24, 40, 100, 136])
        k, s, p = (kh, kw), (pick([1, 1, 2]), pick([1, 1, 2])), (pick([0, kh // 2]), pick([0, kw // 2]))
        h, w = pick([1, 5, 11, 16, 20, 33]), pick([1, 7, 13, 20, 32, 50])
    h, w = max(h, k[0] - 2 * p[0]), max(w, k[1] - 2 * p[1])        # at least one output pixel
    n = pick([1, 1, 2, 3, 5])
    return fam, (cin, cout, k, s, p, h, w, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    import test_gpu_nets as T
    from vse_amd import engine
    ctx = engine.Context(0)
    rng = np.random.default_rng(a.seed)
    bad, fams, t0 = [], {}, time.time()
    for i in range(a.cases):
        fam, c = draw(rng)
        fams[fam] = fams.get(fam, 0) + 1
        try:
            T.test_conv_shapes.__wrapped__(ctx, *c) if hasattr(T.test_conv_shapes, "__wrapped__") else T.test_conv_shapes(ctx, *c)
        except AssertionError as e:
            bad.append((fam, c, "mismatch " + str(e)[:80]))
        except Exception as e:                                      # noqa: BLE001 - report and go on
            bad.append((fam, c, type(e).__name__ + ": " + str(e)[:120]))
    print(f"{a.cases} cases in {time.time() - t0:.0f} s, per family {fams}; failures: {len(bad)}")
    for b in bad:
        print("FAIL", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
