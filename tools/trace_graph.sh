#!/bin/bash
# Usage (GPU box): bash tools/trace_graph.sh <tag> [bench args]   -- kernel trace of the hipGraph-replayed step (real concurrency)
tag=$1; shift
root=$(pwd)
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tg_$tag -o $tag -- python3 $root/bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 3 "$@" > $root/gpurun_out/${tag}_bench.log 2>&1
f=$(find /tmp/tg_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$root/gpurun_out/${tag}_timeline.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
a, b = ad[-3], ad[-2]   # one full replayed step
t0 = int(rows[a]["End_Timestamp"])
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["start_us", "dur_us", "queue", "grid", "wg", "kernel"])
    for r in rows[a + 1:b + 1]:
        w.writerow([round((int(r["Start_Timestamp"]) - t0) / 1e3, 2), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 2),
                    r["Queue_Id"], r["Grid_Size_X"], r["Workgroup_Size_X"], r["Kernel_Name"][:120]])
print("step kernels", b - a, "step us", (int(rows[b]["End_Timestamp"]) - t0) / 1e3)
PY
tail -2 $root/gpurun_out/${tag}_bench.log | cut -c1-200
