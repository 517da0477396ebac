#!/usr/bin/env python3
"""Why does ONE timed block of a secondary bench workload sometimes run at half speed right behind the headline's profiling pass
(bench.py `config.secondary`)?  Reproduces that sequence N times in one process and records, per timed block:
  * wall time, frames/s
  * torch caching-allocator deltas (device allocs / frees, reserved bytes) — a workspace slot or recogniser plan first used INSIDE the
    timed region shows up here (engine.Net._workspace allocates and zero-fills per (plan, slot) on first use)
  * host timestamps of every recogniser call's return (is the slowdown one stall or every step?)
  * GPU clock / power samples from sysfs (hwmon freq1_input / power1_average) taken by a thread every ~3 ms
usage (GPU box): python tools/half_speed_probe.py [--iters 10] [--warm 2] > gpurun_out/half_speed_probe.json
"""
import argparse
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.files = {}
        for pat, key in (("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", "sclk_hz"),
                         ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "power_uw"),
                         ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", "power_uw"),
                         ("/sys/class/drm/card*/device/hwmon/hwmon*/temp1_input", "temp_mc")):
            for p in sorted(glob.glob(pat)):
                self.files.setdefault(key, p)
        self.samples = []
        self.stop = False

    def run(self):
        while not self.stop:
            row = [time.perf_counter()]
            for key in ("sclk_hz", "power_uw", "temp_mc"):
                try:
                    row.append(int(open(self.files[key]).read().strip()))
                except Exception:
                    row.append(None)
            self.samples.append(row)
            time.sleep(0.003)

    def window(self, t0, t1):
        rows = [r for r in self.samples if t0 <= r[0] <= t1]
        out = {"n": len(rows)}
        for i, key in enumerate(("sclk_hz", "power_uw", "temp_mc"), 1):
            vals = [r[i] for r in rows if r[i] is not None]
            if vals:
                out[key] = {"min": min(vals), "max": max(vals), "mean": round(sum(vals) / len(vals), 1)}
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warm", type=int, default=2, help="warm-up steps of the first block (bench.py round 4: 2)")
    ap.add_argument("--blocks", type=int, default=3)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--no-profile-pass", action="store_true", help="skip the headline's roofline pass + vendor GEMM in front of each iteration")
    ap.add_argument("--gc", action="store_true", help="leave Python's cyclic collector ON inside the timed blocks (bench.py turns it off) and log its passes")
    pa = ap.parse_args()
    if pa.gc:
        os.environ["VSE_BENCH_GC"] = "1"
    import gc
    gc_log = []
    gc_t0 = {}

    def gc_cb(phase, info):
        if phase == "start":
            gc_t0[info["generation"]] = time.perf_counter()
        else:
            t0 = gc_t0.pop(info["generation"], None)
            if t0 is not None:
                gc_log.append((t0, info["generation"], round(1e3 * (time.perf_counter() - t0), 2), info.get("collected", 0)))
    gc.callbacks.append(gc_cb)
    sys.argv = ["bench.py"]
    import bench
    args = bench.parse()
    import torch
    from vse_amd import engine, pipeline
    ctx = engine.Context(0)

    def sync():
        torch.cuda.synchronize()

    def log(msg):
        print("[probe]", msg, file=sys.stderr, flush=True)

    W = bench.build_workload(args, ctx, 1, 0, ctx.tdev, sync, log, "server", 1080, 1920, 64)
    _o, dt = W.timed(5, 20)
    log(f"headline: {64 * 20 / dt:.1f} frames/s")
    smp = Sampler()
    log(f"sysfs files: {smp.files}")
    smp.start()
    result = {"headline_fps": round(64 * 20 / dt, 1), "sysfs": smp.files, "iters": []}

    def profile_pass():
        ready = []
        for k in range(W.span):
            maps = W.det_maps()
            db = ctx.db_postprocess(maps, 1080, 1920, **W.pipe.db)
            ready.append((k, [pipeline.sorted_boxes(b[0]) for b in db]))
        W.stage2_recognise(ready)

    def mstats():
        s = torch.cuda.memory_stats()
        return {"device_alloc": s.get("num_device_alloc", 0), "device_free": s.get("num_device_free", 0),
                "reserved_mb": s.get("reserved_bytes.all.current", 0) >> 20, "allocated_mb": s.get("allocated_bytes.all.current", 0) >> 20,
                "alloc_retries": s.get("num_alloc_retries", 0)}

    for it in range(pa.iters):
        rec = {"iter": it, "blocks": []}
        if not pa.no_profile_pass:
            t0 = time.perf_counter()
            bench.roofline(W.pipe, profile_pass, steps_per_call=W.span)
            rec["profile_pass_s"] = round(time.perf_counter() - t0, 3)
        m0 = mstats()
        t0 = time.perf_counter()
        W2 = bench.build_workload(args, ctx, 1, 0, ctx.tdev, sync, log, "server", 2160, 3840, 32)
        rec["build_s"] = round(time.perf_counter() - t0, 3)
        rec["build_alloc"] = {k: mstats()[k] - m0[k] for k in m0}
        stamps = []
        orig_multi, orig_one = W2.pipe.recognize_multi, W2.pipe.recognize

        def rec_multi(*a, **kw):
            r = orig_multi(*a, **kw)
            stamps.append(time.perf_counter())
            return r

        def rec_one(*a, **kw):
            r = orig_one(*a, **kw)
            stamps.append(time.perf_counter())
            return r
        W2.pipe.recognize_multi, W2.pipe.recognize = rec_multi, rec_one
        for b in range(pa.blocks):
            m0 = mstats()
            del stamps[:]
            t0 = time.perf_counter()
            _o2, dtb = W2.timed(pa.warm if b == 0 else 0, pa.steps)
            t1 = time.perf_counter()
            m1 = mstats()
            tstart = t1 - dtb          # (timed() syncs on both sides of the timed run; the warm-up precedes it)
            gaps = [round(1e3 * (s - tstart), 2) for s in stamps if s >= tstart]
            rec["blocks"].append({"fps": round(32 * pa.steps / dtb, 1), "ms": round(1e3 * dtb, 2), "warmup": pa.warm if b == 0 else 0,
                                  "alloc_delta": {k: m1[k] - m0[k] for k in m0}, "rec_call_return_ms": gaps,
                                  "gc_passes": [(round(1e3 * (g[0] - tstart), 1), g[1], g[2], g[3]) for g in gc_log if tstart <= g[0] <= t1],
                                  "gpu": smp.window(tstart, t1)})
        del W2, _o2
        torch.cuda.empty_cache()
        fps = [b["fps"] for b in rec["blocks"]]
        log(f"iter {it}: blocks {fps}  alloc {[b['alloc_delta']['device_alloc'] for b in rec['blocks']]}  gc passes (ms into block, generation, ms, "
            f"collected) {[b['gc_passes'] for b in rec['blocks']]}")
        result["iters"].append(rec)
    smp.stop = True
    allf = [b["fps"] for r in result["iters"] for b in r["blocks"]]
    result["summary"] = {"blocks": len(allf), "min_fps": min(allf), "max_fps": max(allf),
                         "slow_blocks": sum(1 for f in allf if f < 0.8 * max(allf))}
    print(json.dumps(result))


if __name__ == "__main__":
    main()
