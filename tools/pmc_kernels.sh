#!/bin/bash
# Usage (GPU box): bash tools/pmc_kernels.sh <tag> "<counters>" <microbench args...>
# Per-kernel mean of SQ/TCC counters over a tools/microbench.py run -> gpurun_out/<tag>_pmc.txt
tag=$1; ctrs=$2; shift; shift
root=$(pwd); mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pk_$tag -o p -- python3 $root/tools/microbench.py "$@" > /tmp/pk_$tag.log 2>&1
python3 - "$tag" "$root" <<'PY'
import csv, glob, sys, collections
tag, root = sys.argv[1], sys.argv[2]
f = glob.glob(f"/tmp/pk_{tag}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:90], r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[k]["dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
with open(f"{root}/gpurun_out/{tag}_pmc.txt", "w") as out:
    for k, c in acc.items():
        if "at::native" in k[0] or "rocclr" in k[0]:
            continue
        n = len(c["dur_ns"]) // max(1, len(c) - 1)
        line = f"{k[0]} grid={k[1]} wg={k[2]} vgpr={k[3]}+{k[4]} lds={k[5]} n={n} " + " ".join(
            f"{name}={sum(v) / len(v):.4g}" for name, v in sorted(c.items()))
        print(line); out.write(line + "\n")
PY
