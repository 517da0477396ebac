#!/usr/bin/env python3
"""Turn the rocprofv3 outputs merged back under gpurun_out/ into the committed summaries under profiles/.

  gpurun_out/prof_rNN/bench_kernel_stats.csv      -> profiles/rNN_kernel_stats.csv   (verbatim, top rows)
  gpurun_out/pmc_fetch + pmc_write counter CSVs    -> profiles/rNN_traffic.json       (HBM bytes per launch per kernel)
  gpurun_out/pmc_sq counter CSV                    -> profiles/rNN_mfma_busy.json     (matrix-pipe busy fraction per kernel)

MFMA busy: SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over the chip's 1024 SIMDs (32 per
v_mfma_f32_32x32x16_f16, MI355X_MICROARCH.md per-instruction table), so busy fraction = counter / (kernel duration in
ns x 2.4 GHz x 1024); at 1024 FLOP per SIMD-cycle this is also (padded FLOP/s) / 2.5 PFLOP/s.  The wave-state
fractions are the SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY shares of SQ_WAVE_CYCLES (disjoint buckets).

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
    seq = os.path.join(go, f"prof_{tag}_seq", "bench_kernel_stats.csv")
    if os.path.exists(seq):
        shutil.copy(seq, os.path.join(out, f"{tag}_kernel_stats_sequential.csv"))
    sj = os.path.join(go, f"prof_{tag}_seq_bench.json")
    if os.path.exists(sj):
        shutil.copy(sj, os.path.join(out, f"{tag}_bench_sequential_under_rocprof.json"))
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
    def pmc(name):          # gpurun_out/pmc_<name>_<tag> (collect_profiles.sh since round 4), else the un-tagged directory of rounds 1-3
        d = os.path.join(go, f"pmc_{name}_{tag}")
        return d if os.path.isdir(d) else os.path.join(go, f"pmc_{name}")
    f = load(os.path.join(pmc("fetch"), "bench_counter_collection.csv"))
    w = load(os.path.join(pmc("write"), "bench_counter_collection.csv"))
    # counter units per kernel family from the newest calibration run (tools/ubench/counter_calib.hip, tools/counter_calibration.py):
    # known bytes / counter for that family's access shape; the guide's 2.0 / 1.0 for everything that was not calibrated on its own
    import glob
    calib, calib_src = {}, None
    for cpath in sorted(glob.glob(os.path.join(out, "r*_counter_calibration.json")), reverse=True):
        cj = json.load(open(cpath))
        calib, calib_src = cj.get("families", {}), "profiles/" + os.path.basename(cpath)
        dflt = cj.get("default", {})
        break
    else:
        dflt = {}

    def unit(kname):
        fam = calib.get(kname.split("<")[0], {})
        return (fam.get("fetch_factor") or dflt.get("fetch_factor") or 2.0, fam.get("write_factor") or dflt.get("write_factor") or 1.0)
    kernels = {}
    for k in f:
        nf, nw = len(f[k]), len(w.get(k, []))
        fk = sum(f[k]) / nf
        wk = sum(w[k]) / nw if nw else 0.0
        uf, uw = unit(k)
        kernels[k] = {"launches_fetch_pass": nf, "launches_write_pass": nw, "fetch_size_kib_avg": round(fk, 1),
                      "write_size_kib_avg": round(wk, 1), "fetch_unit": uf, "write_unit": uw,
                      "hbm_bytes_per_launch": int(uf * fk * 1024 + uw * wk * 1024)}
    if kernels:
        with open(os.path.join(out, f"{tag}_traffic.json"), "w") as fh:
            json.dump({"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                                 "`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline`; "
                                 "bytes = fetch_unit*FETCH_SIZE*1024 + write_unit*WRITE_SIZE*1024; the units per kernel family come from "
                                 + (calib_src or "the guide (2.0 / 1.0)") + " (known byte counts in each family's access shape: 2.0 / 1.0 for "
                                 "every shape measured, the 16-byte gathers at pixel stride included)",
                       "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_fetch_pass"]))},
                      fh, indent=1)
    sq = os.path.join(pmc("sq"), "bench_counter_collection.csv")
    if os.path.exists(sq):
        cnt = collections.defaultdict(lambda: collections.defaultdict(float))
        dur = collections.defaultdict(float)
        launches = collections.Counter()
        seen = set()
        # launches of >= 100 us on their own: the GRBM counter of a short launch also covers the profiler's own activity
        # around it, so the effective clock is taken from the long ones
        lcnt = collections.defaultdict(lambda: collections.defaultdict(float))
        ldur = collections.defaultdict(float)
        lseen = set()
        for r in csv.DictReader(open(sq)):
            k = short(r["Kernel_Name"])
            cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if d >= 100000:
                lcnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in lseen:
                    lseen.add(r["Dispatch_Id"])
                    ldur[k] += d
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                launches[k] += 1
        total = sum(dur.values())
        ks = {}
        for k in sorted(dur, key=lambda k: -dur[k]):
            c = cnt[k]
            if dur[k] < 0.002 * total:
                continue
            wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            ks[k] = {"launches": launches[k], "total_us": round(dur[k] / 1e3, 1), "share_of_gpu_time": round(dur[k] / total, 4),
                     "mfma_busy_frac": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (dur[k] * 2.4 * 1024), 4),
                     "mfma_mops_f16_per_us": round(c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) / (dur[k] / 1e3), 1),
                     "grbm_gui_active_over_duration": round(c.get("GRBM_GUI_ACTIVE", 0.0) / (dur[k] * 2.4), 3),
                     "wave_wait_any": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 3),
                     "wave_wait_inst_any": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
                     "wave_active_inst_any": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3)}
            if ldur.get(k) and lcnt[k].get("GRBM_GUI_ACTIVE"):
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs (a bandwidth-bound fill kernel reads 8 x 2.41 cycles per ns: the
                # maximum clock); MI355X_MICROARCH.md: effective clock = GRBM_GUI_ACTIVE / wall time
                clk = lcnt[k]["GRBM_GUI_ACTIVE"] / 8.0 / ldur[k]
                ks[k]["effective_clock_ghz"] = round(clk, 3)
                ks[k]["mfma_busy_of_elapsed_shader_cycles"] = round(lcnt[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) /
                                                                    (ldur[k] * clk * 1024), 4)
        conv = [k for k in ks if k.startswith("conv_")]
        tot_busy = sum(cnt[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for k in conv)
        tot_dur = sum(dur[k] for k in conv)
        with open(os.path.join(out, f"{tag}_mfma_busy.json"), "w") as fh:
            json.dump({"method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY "
                                 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE over "
                                 "`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline`; "
                                 "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (duration_ns * 2.4 * 1024 SIMDs), i.e. against the peak at "
                                 "the 2.4 GHz maximum clock; effective_clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / duration over launches "
                                 ">= 100 us (DVFS: the chip clocks to its power budget), mfma_busy_of_elapsed_shader_cycles = the "
                                 "same busy cycles against the cycles that actually elapsed",
                       "all_conv_kernels_mfma_busy_frac": round(tot_busy / (tot_dur * 2.4 * 1024), 4) if tot_dur else None,
                       "kernels": ks}, fh, indent=1)
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
