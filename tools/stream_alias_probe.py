#!/usr/bin/env python3
"""Do two HIP streams really run concurrently?  torch hands `torch.cuda.Stream()` objects out of a round-robin pool of 32 per priority,
and the HIP runtime multiplexes all streams of one priority over a handful of hardware queues (GPU_MAX_HW_QUEUES, default 4): two
streams that share a hardware queue execute IN ORDER, whatever the program says.  This probe
  1. draws streams from torch's pool and measures, pair by pair, whether a short kernel on stream B can overtake a long spin on
     stream A (concurrent) or has to wait for it (aliased), against the null stream and against each other;
  2. times the 4K secondary workload of bench.py with detector streams that are (a) all independent of the main stream and
     (b) one of them aliased with the main stream — the configuration every second workload of a process used to get.
usage (GPU box): python tools/stream_alias_probe.py > gpurun_out/stream_alias_probe.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def concurrent(a, b, x, spin=4_000_000):
    """True when a tiny kernel on stream b finishes while a long spin on stream a is still running."""
    torch.cuda.synchronize()
    a0, a1, b1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(a):
        a0.record()
        torch.cuda._sleep(spin)
        a1.record()
    with torch.cuda.stream(b):
        x.add_(1)
        b1.record()
    torch.cuda.synchronize()
    return a0.elapsed_time(b1) < 0.5 * a0.elapsed_time(a1), round(a0.elapsed_time(a1), 3), round(a0.elapsed_time(b1), 3)


def main():
    dev = torch.device("cuda", 0)
    x = torch.zeros(64, device=dev)
    null = torch.cuda.default_stream(dev)
    out = {"env_GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "normal": [], "high": []}
    normal = [torch.cuda.Stream(device=dev) for _ in range(12)]
    high = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(8)]
    for name, pool in (("normal", normal), ("high", high)):
        for i, s in enumerate(pool):
            vs_null = concurrent(null, s, x)
            vs_first = concurrent(pool[0], s, x) if i else (None, None, None)
            out[name].append({"pool_index": i, "concurrent_with_null": vs_null[0], "spin_ms": vs_null[1], "tiny_done_ms": vs_null[2],
                              "concurrent_with_pool0": vs_first[0]})
    # alias classes among the normal-priority streams (which pool entries share a hardware queue)
    classes = []
    for i, s in enumerate(normal):
        for c in classes:
            if not concurrent(normal[c[0]], s, x)[0]:
                c.append(i)
                break
        else:
            classes.append([i])
    out["normal_alias_classes"] = classes
    out["normal_aliased_with_null"] = [i for i, e in enumerate(out["normal"]) if not e["concurrent_with_null"]]
    out["high_vs_normal0_concurrent"] = [concurrent(normal[0], s, x)[0] for s in high]
    print(json.dumps(out), flush=True)
    if "--no-workload" in sys.argv:
        return
    # ---- the 4K workload with chosen detector streams --------------------------------------------------------------------------
    sys.argv = ["bench.py"]
    import bench
    from vse_amd import engine
    args = bench.parse()
    ctx = engine.Context(0)
    good = [s for i, s in enumerate(normal) if out["normal"][i]["concurrent_with_null"]]
    bad = [s for i, s in enumerate(normal) if not out["normal"][i]["concurrent_with_null"]]
    # two independent streams that do not alias each other either
    pair_good = [good[0]] + [s for s in good[1:] if concurrent(good[0], s, x)[0]][:1]
    res = {}
    real_stream = torch.cuda.Stream
    for label, pair in (("independent", pair_good), ("one_aliased_with_main", [pair_good[0]] + bad[:1]), ("independent_again", pair_good)):
        if len(pair) < 2:
            res[label] = "no such pair on this box"
            continue
        it = iter(pair)

        def fake(device=None, priority=0, _it=it):
            if priority == 0:
                try:
                    return next(_it)
                except StopIteration:
                    pass
            return real_stream(device=device, priority=priority)
        torch.cuda.Stream = fake
        try:
            W2 = bench.build_workload(args, ctx, 1, 0, ctx.tdev, torch.cuda.synchronize, lambda m: None, "server", 2160, 3840, 32)
        finally:
            torch.cuda.Stream = real_stream
        fps = []
        for b in range(3):
            _o, dt = W2.timed(4 if b == 0 else 0, 8)
            fps.append(round(32 * 8 / dt, 1))
        res[label] = fps
        del W2, _o
        torch.cuda.empty_cache()
        print(f"[probe] {label}: {fps}", file=sys.stderr, flush=True)
    print(json.dumps({"workload_4k_batch32_fps": res}), flush=True)


if __name__ == "__main__":
    main()
