#!/usr/bin/env python3
"""Print a compact per-kernel table from a rocprofv3 --kernel-trace --stats kernel_stats.csv."""
import csv
import re
import sys


def short(n):
    n = n.replace("focal_gemm_kernel", "GEMM")
    m = re.match(r"_Z\d+GEMMI(.*)EEv10GemmParams", n)
    return n[:120]


rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / 1e6:.2f} ms over {steps:g} steps = {tot / 1e6 / steps:.3f} ms/step")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step {float(r['Percentage']):5.1f}% calls/step={float(r['Calls']) / steps:6.1f} "
          f"avg={float(r['AverageNs']) / 1e3:8.1f}us  {short(r['Name'])}")
