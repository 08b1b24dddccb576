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
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcs_${tag}_$c -o $c -- python3 $root/bench.py --no-graph --no-cpu-baseline --no-roofline --steps $STEPS --warmup $WARM "$@" > /tmp/pmcs_${tag}_$c.log 2>&1
done
python3 - "$tag" "$root" $STEPS $WARM <<'PY' | tee $root/gpurun_out/${tag}_step_traffic.txt
import csv, glob, sys, collections, re
tag, root, steps, warm = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
def fam(n):
    n = re.sub(r"\(Gemm(Epi|Pro)\)", "", n)
    if "gemm_ring" in n: return "gemm_ring"
    if "gemm_pipe" in n: return "gemm_pipe"
    if "focal_dw_ring" in n or "focal_dw_group" in n: return "gemm dW"
    if "mlp_bwd" in n: return "mlp_bwd"
    if "mlp_fwd" in n: return "mlp_fwd"
    if "focal_gemm_kernel" in n:
        if "Lb1ELb1" in n or "true, true" in n: return "gemm dW"
        return "gemm fwd/dX 64x64"
    for k in ("ln_bwd", "ln_fwd", "window_attn_bwd", "window_attn_fwd", "patch_embed", "fft_realpack", "adamw", "mask_cast", "copyBuffer", "fillBuffer"):
        if k in n: return k
    return "other"
tot = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0, 0.0])  # fetch KB, write KB, launches, MFMA-busy cycles, duration ns
for ci, c in enumerate(("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES")):
    fs = glob.glob(f"/tmp/pmcs_{tag}_{c}/**/*counter_collection.csv", recursive=True)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != c: continue
        e = tot[fam(r["Kernel_Name"])]
        if ci < 2:
            e[ci] += float(r["Counter_Value"])
        else:
            e[3] += float(r["Counter_Value"])
            e[4] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if ci == 0: e[2] += 1
n = steps + warm
print(f"Per kernel family, whole run of {n} eager steps (+ bench.py's two arena-building steps): fabric traffic = (2*FETCH_SIZE + WRITE_SIZE) KB,")
print("serialised kernel time and MFMA-busy cycles from the SQ_VALU_MFMA_BUSY_CYCLES pass; GB/s against the 8 TB/s HBM peak, MFMA busy =")
print("busy cycles / (time x 2.4 GHz x 1024 SIMDs).  Traffic includes infinity-cache hits (the counters sit at the L2 <-> fabric boundary).")
gt, tt = 0.0, 0.0
for k, (f, w, cnt, busy, dur) in sorted(tot.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1])):
    gb = (2 * f + w) * 1024 / 1e9
    gt += gb
    tt += dur
    gbs = gb / (dur * 1e-9) if dur else 0.0
    mf = busy / (dur * 1e-9 * 2.4e9 * 1024) if dur else 0.0
    print(f"  {k:22s} read {2*f*1024/1e9:8.3f} GB  write {w*1024/1e9:8.3f} GB  time {dur/1e6:8.3f} ms  {gbs:7.0f} GB/s ({gbs/8000*100:4.1f} % of peak)  MFMA busy {mf*100:5.1f} %  launches {cnt:6d}")
print(f"  TOTAL {gt:.3f} GB in {tt/1e6:.3f} ms of kernel time = {gt/(tt*1e-9):.0f} GB/s")
print(f"  per step: {gt / (n + 2):.3f} GB ({n} bench steps + bench.py's two arena-building steps = {n + 2} steps in the total)")
# per traced kernel / op: what bench.py's `roofline.traffic` quotes (HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE, KB -> B),
# keyed like bench.py's StepTracer names its groups (kernel template name, or the focal_amd.ops function for the generic ones)
import hashlib, json, os
def key_of(n):
    m = re.sub(r"\(Gemm(Epi|Pro)\)", "", n)
    for k in ("focal_dw_group_kernel", "focal_dw_ring_kernel", "ln_bwd_kernel", "mlp_bwd_kernel"):
        if k in m: return k
    table = (("mlp_fwd_kernel", "mlp_fwd"), ("ln_fwd_kernel", "layernorm_fwd"), ("window_attn_fwd", "window_attn_fwd"), ("window_attn_bwd", "window_attn_bwd"),
             ("gru_seq_fwd", "gru_seq_fwd"), ("gru_seq_bwd", "gru_seq_bwd"), ("bn_bwd_apply", "bn_act_bwd"), ("bn_bwd_reduce", "bn_act_bwd"), ("bn_partial", "bn_stats"),
             ("bn_act_fwd", "bn_act_fwd"), ("permute_pack", "permute_pack"), ("permute_unpack", "permute_unpack_add"), ("conv_pack_bwd", "conv_pack_bwd"),
             ("conv_in_bwd_weight", "conv_in_bwd_weight"), ("conv_in_fwd", "conv_in_fwd"), ("fft_realpack", "fft_realpack"), ("patch_embed", "pad_patch_embed_ln"),
             ("adamw_kernel", "adamw_multi"), ("mask_cast", "mask_cast"))
    for pat, k in table:
        if pat in m: return k
    if "focal_gemm_pipe_kernel" in m:
        trb = ("Lb1E" in m.split("focal_gemm_pipe_kernel")[1][:40]) if m.startswith("_Z") else (", true," in m.split("focal_gemm_pipe_kernel")[1][:60])
        return "linear_bwd_data" if trb else "linear_fwd"
    if "focal_gemm_kernel" in m:
        if "Lb1ELb1" in m or "true, true" in m: return "focal_gemm_kernel"   # register-staged weight gradient
        if "Lb0ELb1" in m or "false, true" in m: return "linear_bwd_data"
        return "linear_fwd"
    return None
grp = collections.defaultdict(lambda: [0.0, 0.0, 0])
for ci, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    fs = glob.glob(f"/tmp/pmcs_{tag}_{c}/**/*counter_collection.csv", recursive=True)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != c: continue
        key = key_of(r["Kernel_Name"])
        if key is None: continue
        g = grp[key]
        g[ci] += float(r["Counter_Value"])
        if ci == 0: g[2] += 1
out = {k: {"launches": v[2], "hbm_bytes_per_launch": (2 * v[0] + v[1]) * 1024 / max(v[2], 1),
           "fetch_bytes_per_launch_corrected": 2 * v[0] * 1024 / max(v[2], 1), "write_bytes_per_launch": v[1] * 1024 / max(v[2], 1)}
       for k, v in grp.items()}
lib = os.path.join(root, "focal_amd", "libfocal_hip.so")
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over an eager bench.py run; bytes = "
                   "2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts half of a wide coalesced read, MI355X_MICROARCH.md HBM section); "
                   "the counters sit at the L2 <-> fabric boundary: Infinity-Cache hits are included.  Keys = bench.py StepTracer group names; "
                   "the linear_fwd / linear_bwd_data keys also hold the DeepSense convolutions that run on the same GEMM templates",
           "lib_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16], "groups": out},
          open(f"{root}/gpurun_out/{tag}_pmc_groups.json", "w"), indent=1)
PY
