#!/usr/bin/env python3
"""What the recogniser costs the streamed headline: bench.py's workload with the recognition stage replaced by empty results (detector +
DB post-processing + box ordering only), next to the full step.  usage: python tools/bench_det_only.py [--steps 20]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def main():
    sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:]]
    args = bench.parse()
    import torch
    from vse_amd import engine
    ctx = engine.Context(0)
    args.dist_on, args.dist_backend = False, "none"

    def sync():
        torch.cuda.synchronize()
    W = bench.build_workload(args, ctx, 1, 0, "cpu", sync, lambda m: None, args.models, args.height, args.width, args.batch)
    for label in ("full", "det+db only", "full", "det+db only"):
        if label != "full":
            W.pipe.recognize_multi_launch = lambda fl, bl: bl
            W.pipe.recognize_multi_collect = lambda bl: [[[("", 0.0)] * len(b) for b in boxes] for boxes in bl]
        else:
            W.pipe.__dict__.pop("recognize_multi_launch", None)
            W.pipe.__dict__.pop("recognize_multi_collect", None)
        out, dt = W.timed(args.warmup, args.steps)
        print(f"{label:12s}: {args.batch * args.steps / dt:8.1f} frames/s  {1e3 * dt / args.steps:7.3f} ms / step", flush=True)


if __name__ == "__main__":
    main()
