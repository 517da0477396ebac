#!/usr/bin/env python3
"""GPU idle time inside the timed steps of a rocprofv3 --kernel-trace CSV: union of kernel intervals vs span.
usage: python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]   (skips the leading warm-up part of the run)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]) for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
iv = iv[int(len(iv) * skip):]
busy, gaps = 0, []
cs, ce = iv[0][0], iv[0][1]
last = iv[0][2]
for s, e, n in iv[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append((s - ce, last[:24] + ' -> ' + n))
        cs, ce = s, e
    else:
        ce = max(ce, e)
    last = n
busy += ce - cs
span = iv[-1][1] - iv[0][0]
print(f"kernels {len(iv)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {(span - busy) / 1e6:.2f} ms ({100 * (span - busy) / span:.1f} %)")
small = [g for g, _ in gaps if g < 100000]
print(f"gaps {len(gaps)}: < 100 us: {len(small)} summing {sum(small) / 1e6:.2f} ms (avg {sum(small) / max(1, len(small)) / 1e3:.1f} us); "
      f"largest (us, previous -> next kernel): {[(round(g / 1e3), n[:60]) for g, n in sorted(gaps, reverse=True)[:6]]}")
ksum = sum(e - s for s, e, _ in iv)
print(f"sum of kernel durations {ksum / 1e6:.2f} ms -> average concurrency {ksum / busy:.2f}")
