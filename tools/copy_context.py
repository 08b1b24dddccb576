"""Neighbours of the small __amd_rocclr_copyBuffer launches in a rocprofv3 kernel trace: what runs right before / after them?"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ctx = collections.Counter()
for i, r in enumerate(rows):
    if "copyBuffer" in r["Kernel_Name"]:
        prev = rows[i - 1]["Kernel_Name"][:70] if i else "-"
        nxt = rows[i + 1]["Kernel_Name"][:70] if i + 1 < len(rows) else "-"
        ctx[(r["Grid_Size_X"], prev, nxt)] += 1
for k, v in ctx.most_common(30):
    print(v, k)
