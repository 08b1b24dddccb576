#!/bin/bash
# Usage (GPU box): bash tools/pmc_step_traffic.sh <tag> [bench args]
# HBM traffic of a whole pretraining step, per kernel family: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
# --kernel-trace only) over an eager bench run of STEPS steps; bytes = (2*FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction,
# MI355X_MICROARCH.md HBM section).  Result: gpurun_out/<tag>_step_traffic.txt
tag=$1; shift
root=$(pwd)
STEPS=4; WARM=2
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcs_${tag}_$c -o $c -- python3 $root/bench.py --no-graph --no-cpu-baseline --no-roofline --steps $STEPS --warmup $WARM "$@" > /tmp/pmcs_${tag}_$c.log 2>&1
done
python3 - "$tag" "$root" $STEPS $WARM <<'PY' | tee $root/gpurun_out/${1}_step_traffic.txt
import csv, glob, sys, collections, re
tag, root, steps, warm = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
def fam(n):
    n = re.sub(r"\(Gemm(Epi|Pro)\)", "", n)
    if "gemm_pipe" in n: return "gemm_pipe"
    if "focal_gemm_kernel" in n:
        if "Lb1ELb1" in n or "true, true" in n: return "gemm dW"
        return "gemm fwd/dX 64x64"
    for k in ("ln_bwd", "ln_fwd", "window_attn_bwd", "window_attn_fwd", "patch_embed", "fft_realpack", "adamw", "mask_cast", "copyBuffer", "fillBuffer"):
        if k in n: return k
    return "other"
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for ci, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    fs = glob.glob(f"/tmp/pmcs_{tag}_{c}/**/*counter_collection.csv", recursive=True)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != c: continue
        e = tot[fam(r["Kernel_Name"])]
        e[ci] += float(r["Counter_Value"])
        if ci == 0: e[2] += 1
n = steps + warm
print(f"HBM traffic per step (mean over {n} eager steps incl. warmup; KB counters -> GB):")
gt = 0.0
for k, (f, w, cnt) in sorted(tot.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1])):
    gb = (2 * f + w) * 1024 / n / 1e9
    gt += gb
    print(f"  {k:22s} read {2*f*1024/n/1e9:7.3f} GB  write {w*1024/n/1e9:7.3f} GB  total {gb:7.3f} GB   launches/step {cnt/n:6.1f}")
print(f"  TOTAL {gt:.3f} GB/step")
PY
