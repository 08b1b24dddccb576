#!/usr/bin/env python3
"""Per-INSTANCE kernel table from a rocprofv3 --kernel-trace CSV (one row per dispatch).

  python tools/instance_table.py <kernel_trace.csv> <steps> [out.csv] [--filter substr]

rocprofv3's --stats summary averages a kernel TEMPLATE over every shape it is launched with (the weight-gradient GEMM
template runs 74 different linear layers per step); this groups the dispatches by (kernel name, grid size, workgroup size,
LDS bytes) -- one row per launch shape -- so that bytes / duration can be formed per instance.  `steps` = number of steps the
trace covers (warm-up included) -> calls per step.
"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(Gemm(Epi|Pro)\)", "", n)
    n = n.replace("focal_gemm_kernel", "GEMM").replace("focal_gemm_pipe_kernel", "PIPE")
    return n[:110]


def load(path):
    g = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        key = (r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1),
               int(r["Workgroup_Size_X"]), int(r.get("LDS_Block_Size", 0) or 0))
        g[key].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
    return g


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    filt = None
    if "--filter" in sys.argv:
        filt = sys.argv[sys.argv.index("--filter") + 1]
        args = [a for a in args if a != filt]
    path, steps = args[0], float(args[1])
    out = args[2] if len(args) > 2 else None
    g = load(path)
    rows = []
    for (name, grid, wg, lds), d in g.items():
        if filt and filt not in name:
            continue
        d.sort()
        rows.append(dict(kernel=short(name), grid_threads=grid, workgroups=grid // max(wg, 1), wg=wg, lds=lds, calls=len(d),
                         calls_per_step=round(len(d) / steps, 2), avg_us=round(sum(d) / len(d), 2), med_us=round(d[len(d) // 2], 2),
                         min_us=round(d[0], 2), max_us=round(d[-1], 2), ms_per_step=round(sum(d) / steps / 1e3, 4)))
    rows.sort(key=lambda r: -r["ms_per_step"])
    tot = sum(r["ms_per_step"] for r in rows)
    print(f"# {len(rows)} launch shapes, {tot:.3f} ms of kernel time per step ({steps:g} steps)")
    for r in rows[:60]:
        print(f"{r['ms_per_step']:8.4f} ms/step  calls/step {r['calls_per_step']:6.2f}  avg {r['avg_us']:8.2f} us  med {r['med_us']:8.2f}  "
              f"wgs {r['workgroups']:6d} x {r['wg']:4d}  {r['kernel']}")
    if out:
        with open(out, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)


if __name__ == "__main__":
    main()
