#!/bin/bash
root=$(pwd); mkdir -p $root/gpurun_out
python3 tools/mb_attn_branch.py 20 > $root/gpurun_out/r4_branch_mb.txt 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pb_$i -o p -- python3 $root/tools/mb_attn_branch.py 4 > /tmp/pb_$i.log 2>&1
done
python3 - "$root" <<'PY' | tee $root/gpurun_out/r4_branch_pmc.txt
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pb_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn" not in n and "PIPE" not in n and "gemm_pipe" not in n: continue
        k = (n[:70], r["Grid_Size"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, c in sorted(acc.items()):
    print(k[0], "grid", k[1])
    for name, v in sorted(c.items()):
        print(f"    {name:32s} {sum(v) / len(v):14.5g}")
PY
