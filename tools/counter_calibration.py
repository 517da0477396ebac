#!/usr/bin/env python3
"""Reduce a run of tools/ubench/counter_calib.hip under rocprofv3 to profiles/<tag>_counter_calibration.json.

On the GPU box (tools/run_counter_calibration.sh does this):
  counter_calib                                          > gpurun_out/calib_<tag>/known.json
  rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- counter_calib     -> gpurun_out/calib_<tag>/fetch/calib_counter_collection.csv
  rocprofv3 --pmc WRITE_SIZE --kernel-trace ... -- counter_calib     -> gpurun_out/calib_<tag>/write/calib_counter_collection.csv
Here:  python tools/counter_calibration.py r05

factor = known bytes / (counter KiB x 1024): what a FETCH_SIZE / WRITE_SIZE reading of a kernel with that access shape has to be
multiplied by.  MI355X_MICROARCH.md (HBM) establishes 2.0 for wide coalesced reads only; tools/summarize_profiles.py looks the factor of
a kernel family up in the newest profiles/*_counter_calibration.json (FAMILY below) and falls back to that 2.0 / 1.0.
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# product kernel family -> the calibration kernels with its access shape: (read UNIT = the same lanes / strides without a neighbourhood,
# every byte read exactly once; the neighbourhood form whose excess over the unit is real re-fetching; the store shape)
FAMILY = {
    "conv_dwpw_kernel": ("calib_pix_read", "calib_pix_read3x3", "calib_pix_write"),
    "conv_pw_kernel": ("calib_pix_read", None, "calib_pix_write"),
    "chain_pw2_kernel": ("calib_pix_read", None, "calib_pix_write"),
    "dwconv_row_kernel": ("calib_pix_read", "calib_dwrow_read", "calib_stream_write"),
    "dwconv_kernel": ("calib_pix_read", "calib_dwrow_read", "calib_stream_write"),
}


def short(name):
    """Kernel name without return type and argument list.  A stand-alone HIP binary's names arrive mangled, and the binutils c++filt of
    the image does not know _Float16 (DF16_): the few forms used here (_Z<len><name>[I(Li<int>E|Lb<0/1>E)+E]...) are decoded directly."""
    if name.startswith("_Z"):
        import re
        m = re.match(r"_Z(\d+)", name)
        n = int(m.group(1))
        base, rest = name[m.end():m.end() + n], name[m.end() + n:]
        if rest.startswith("I"):
            args, rest = [], rest[1:]
            while True:
                a = re.match(r"Li(\d+)E|Lb([01])E", rest)
                if not a:
                    break
                args.append(a.group(1) if a.group(1) is not None else ("true" if a.group(2) == "1" else "false"))
                rest = rest[a.end():]
            base += "<" + ", ".join(args) + ">"
        return base
    return name.replace("void ", "").split("(")[0]


def load(path):
    agg = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    if not os.path.exists(path):
        return agg, dur
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k].append(float(r["Counter_Value"]))
        if "End_Timestamp" in r:
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return agg, dur


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    d = os.path.join(ROOT, "gpurun_out", f"calib_{tag}")
    known = json.load(open(os.path.join(d, "known.json")))["kernels"]
    fetch, fdur = load(os.path.join(d, "fetch", "calib_counter_collection.csv"))
    write, _ = load(os.path.join(d, "write", "calib_counter_collection.csv"))
    out = {}
    for name, k in known.items():
        if name == "_end":
            continue
        f = fetch.get(name, [])
        w = write.get(name, [])
        # the first launch of a kernel also pays for cold instruction / constant fetches: take the median launch
        fk = sorted(f)[len(f) // 2] if f else None
        wk = sorted(w)[len(w) // 2] if w else None
        e = {"known_bytes": int(k["known_bytes"]), "kind": k["kind"], "fetch_size_kib": fk, "write_size_kib": wk}
        if fdur.get(name):
            us = sorted(fdur[name])[len(fdur[name]) // 2] / 1e3
            e["duration_us_under_pmc"] = round(us, 1)
            e["known_gbs"] = round(k["known_bytes"] / us / 1e3, 1)
        if k["kind"] == "read" and fk:
            e["fetch_factor"] = round(k["known_bytes"] / (fk * 1024), 3)
        if k["kind"] == "write" and wk:
            e["write_factor"] = round(k["known_bytes"] / (wk * 1024), 3)
        out[name] = e

    def fam(prefix, key, xcd=None):
        vals = [v[key] for n, v in out.items() if n.split("<")[0] == prefix and key in v
                and (xcd is None or n.endswith(", true>" if xcd else ", false>"))]
        return round(sum(vals) / len(vals), 3) if vals else None
    unit_f = out.get("calib_stream_read", {}).get("fetch_factor")
    families = {}
    for name, (unit, nbhd, wr) in FAMILY.items():
        e = {"fetch_factor": fam(unit, "fetch_factor"), "fetch_unit_from": unit, "write_factor": fam(wr, "write_factor"), "write_from": wr}
        if nbhd and e["fetch_factor"]:
            # bytes actually fetched (counter x unit factor) over the tensor read once: > 1 = rows re-fetched from HBM by several L2s / evicted
            for xcd, key in ((False, "refetch_plain_block_order"), (True, "refetch_xcd_contiguous_order")):
                f = fam(nbhd, "fetch_factor", xcd)
                e[key] = round(e["fetch_factor"] / f, 2) if f else None
            e["neighbourhood_from"] = nbhd
        families[name] = e
    res = {"method": "tools/ubench/counter_calib.hip under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); every kernel reads "
                     "or writes a known byte count (tensors >= 0.5 GB, beyond the Infinity Cache) with the access shape of one kernel "
                     "family; factor = known bytes / (counter KiB * 1024), median of three launches.  fetch_factor of a family = the factor "
                     "of its lanes / strides WITHOUT a neighbourhood (every byte read once): the counter's unit for that shape.  The 3x3 / "
                     "row-window kernels touch the same tensor through their window: what FETCH_SIZE x unit reports above the tensor size "
                     "is real re-fetching (refetch_*), with the plain blockIdx order and with the product's XCD-contiguous order",
           "kernels": out, "families": families,
           "default": {"fetch_factor": unit_f, "write_factor": out.get("calib_stream_write", {}).get("write_factor")}}
    path = os.path.join(ROOT, "profiles", f"{tag}_counter_calibration.json")
    with open(path, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))
    print("wrote", path)


if __name__ == "__main__":
    main()
