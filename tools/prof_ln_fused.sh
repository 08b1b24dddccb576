#!/bin/bash
# kernel durations of tools/mb_ln_fused.py under rocprofv3, per launch shape: the 64 x 256 ring form (FOCAL_LAB_LN_BWD_RING256=1), the pipe form, the two launches
root=$(cd "$(dirname "$0")/.." && pwd)
for sel in ring256 pipe; do
  cd /tmp && export TMPDIR=/tmp
  if [ $sel == ring256 ]; then export FOCAL_LAB_LN_BWD_RING256=1; else unset FOCAL_LAB_LN_BWD_RING256; fi
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_lnf_$sel -o t -- python3 $root/tools/mb_ln_fused.py > /tmp/prof_lnf_$sel.log 2>&1
  cd $root
  echo "== fused form: $sel"
  python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/prof_lnf_$sel/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if not any(s in k for s in ("focal_gemm", "ln_bwd")): continue
    key = (k[:95], r["Grid_Size_X"], r["Workgroup_Size_X"])
    acc.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g, w), v in acc.items():
    v = sorted(v)
    print(f"{k:95s} grid {g:>7s} x {w:>4s}  n {len(v):3d}  median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f}")
PY
done
