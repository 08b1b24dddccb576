#!/bin/bash
# Usage (GPU box): bash tools/pmc_roofline.sh <tag>
# HBM traffic of the dominant kernel as bench.py's roofline leg launches it: two rocprofv3 --pmc passes (FETCH_SIZE,
# WRITE_SIZE; counters only with --kernel-trace, per MI355X_MICROARCH.md), aggregated into gpurun_out/<tag>_pmc_roofline_kernel.json
tag=$1
root=$(pwd)
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${tag}_$c -o $c -- python3 $root/bench.py --roofline-only --roofline-iters 20 > /tmp/pmc_${tag}_$c.log 2>&1
done
python3 - "$tag" "$root" <<'PY'
import csv, glob, json, sys
tag, root = sys.argv[1], sys.argv[2]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == c and "focal_gemm_kernel" in r["Kernel_Name"] and "true, true, 0, 0, 5" in r["Kernel_Name"].replace("(GemmEpi)", "").replace("(GemmPro)", "")]
    if not vals:  # mangled names
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                if r["Counter_Name"] == c and "focal_gemm_kernel" in r["Kernel_Name"] and "Lb1ELb1ELi0ELi0ELi5E" in r["Kernel_Name"]]
    out[c] = (sum(vals) / max(len(vals), 1), len(vals))
fetch_kb, n = out["FETCH_SIZE"]
write_kb, _ = out["WRITE_SIZE"]
res = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python3 bench.py "
               "--roofline-only --roofline-iters 20`; kernel = the dW GEMM bench.py times live. bytes = (2*FETCH_SIZE + "
               "WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts half of wide streaming reads (MI355X_MICROARCH.md, HBM section); "
               "WRITE_SIZE is the fp32 atomic volume.",
       "launches": n, "mean_FETCH_SIZE_KB": fetch_kb, "mean_WRITE_SIZE_KB": write_kb,
       "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024)}
json.dump(res, open(f"{root}/gpurun_out/{tag}_pmc_roofline_kernel.json", "w"), indent=1)
print(json.dumps(res))
PY
