#!/usr/bin/env python3
"""Turn the rocprofv3 outputs merged back under gpurun_out/ into the committed summaries under profiles/.

  gpurun_out/prof_rNN/bench_kernel_stats.csv      -> profiles/rNN_kernel_stats.csv   (verbatim, top rows)
  gpurun_out/pmc_fetch + pmc_write counter CSVs    -> profiles/rNN_traffic.json       (HBM bytes per launch per kernel)

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are collected in SEPARATE
--pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports exactly half of the bytes of wide coalesced reads, so
bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (WRITE_SIZE is uncalibrated on this chip; it is kept as reported).
usage: python tools/summarize_profiles.py r01
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    go = os.path.join(ROOT, "gpurun_out")
    stats = os.path.join(go, f"prof_{tag}", "bench_kernel_stats.csv")
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))
    bj = os.path.join(go, f"prof_{tag}_bench.json")
    if os.path.exists(bj):
        shutil.copy(bj, os.path.join(out, f"{tag}_bench_under_rocprof.json"))

    def load(path):
        agg = collections.defaultdict(list)
        if not os.path.exists(path):
            return agg
        for r in csv.DictReader(open(path)):
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        return agg
    f = load(os.path.join(go, "pmc_fetch", "bench_counter_collection.csv"))
    w = load(os.path.join(go, "pmc_write", "bench_counter_collection.csv"))
    kernels = {}
    for k in f:
        nf, nw = len(f[k]), len(w.get(k, []))
        fk = sum(f[k]) / nf
        wk = sum(w[k]) / nw if nw else 0.0
        kernels[k] = {"launches_fetch_pass": nf, "launches_write_pass": nw, "fetch_size_kib_avg": round(fk, 1),
                      "write_size_kib_avg": round(wk, 1),
                      "hbm_bytes_per_launch": int(2 * fk * 1024 + wk * 1024)}
    if kernels:
        with open(os.path.join(out, f"{tag}_traffic.json"), "w") as fh:
            json.dump({"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                                 "`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline`; "
                                 "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE half-count correction)",
                       "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_fetch_pass"]))},
                      fh, indent=1)
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
