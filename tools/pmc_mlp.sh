#!/bin/bash
# Usage (GPU box): bash tools/pmc_mlp.sh <tag>   -> gpurun_out/<tag>_pmc.txt
# SQ counters of the fused MLP kernels (separate rocprofv3 --pmc passes, --kernel-trace only), per kernel and launch shape.
tag=$1
root=$(pwd); mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pm_${tag}_$i -o p -- python3 $root/tools/mb_mlp.py 6 > /tmp/pm_${tag}_$i.log 2>&1
done
python3 - "$tag" "$root" <<'PY' | tee $root/gpurun_out/${1}_pmc.txt
import csv, glob, sys, collections
tag, root = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/pm_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mlp_" not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"][:60], r["Grid_Size"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, c in sorted(acc.items()):
    print(k[0], "grid", k[1])
    for name, v in sorted(c.items()):
        print(f"    {name:32s} {sum(v) / len(v):14.5g}")
PY
