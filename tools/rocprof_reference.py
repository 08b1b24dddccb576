#!/usr/bin/env python3
"""Build the committed rocprofv3 reference of one workload from the passes tools/profile_round.sh takes on the GPU box.

  python tools/rocprof_reference.py <tag> <model> <dataset> <steps_in_run> <skip_steps> <trace_dir> <pmc_dir_prefix> <pmc_steps> [<trace_dump.json>]

Inputs: `rocprofv3 --kernel-trace --stats -M` (mangled names) of an eager bench.py run, the --pmc FETCH_SIZE / WRITE_SIZE /
SQ_VALU_MFMA_BUSY_CYCLES passes of the same command (separate passes, --kernel-trace only; bytes = 2 x FETCH_SIZE + WRITE_SIZE KB:
the gfx950 correction of MI355X_MICROARCH.md's HBM section), and bench.py --trace-dump of the same box.
Outputs under gpurun_out/: <tag>_reference.json (what bench.py's roofline is checked against and quotes `traffic` from; keyed by
tools/kernel_names.short_kernel_name), <tag>_step_traffic.txt (per family), <tag>_trace_vs_rocprof.txt (the library's launch trace
next to rocprofv3, kernel by kernel).
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short_kernel_name  # noqa: E402


def fam(n):
    if n.startswith("focal_gemm_ring"): return "gemm_ring"
    if n.startswith("focal_gemm_pipe"): return "gemm_pipe"
    if n.startswith(("focal_dw_ring", "focal_dw_group", "focal_dw_tail_group")): return "gemm dW"
    if n.startswith("focal_gemm_kernel"):
        return "gemm dW" if ", true, true," in n else "gemm fwd/dX 64x64"
    for k in ("mlp_bwd", "mlp_fwd", "swin_attn_branch_bwd", "swin_attn_branch_fwd", "ln_bwd", "ln_fwd", "window_attn_bwd", "window_attn_fwd", "patch_embed", "fft_", "adamw", "mask_cast",
              "gru_seq_bwd", "gru_seq_fwd", "bn_bwd", "bn_partial", "bn_act_fwd", "conv_in", "copyBuffer", "fillBuffer"):
        if k in n: return k
    return "other"


def main():
    tag, model, dataset, steps, skip, trace_dir, pmc_prefix, pmc_steps = sys.argv[1:9]
    steps, skip, pmc_steps = int(steps), int(skip), int(pmc_steps)
    dump = sys.argv[9] if len(sys.argv) > 9 else None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    # ---- kernel trace: per kernel, in dispatch order; the first skip/steps of each kernel's launches (arena building + warm-up) dropped
    tf = glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(tf)):
        per[short_kernel_name(r["Kernel_Name"])].append((float(r["Start_Timestamp"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
    kernels, total_us = {}, 0.0
    for k, v in per.items():
        v.sort()
        drop = int(round(len(v) * skip / steps))
        kept = [d for _, d in v[drop:]] or [d for _, d in v]
        n_steps = steps - skip if drop else steps
        kernels[k] = {"calls_per_step": len(kept) / n_steps, "avg_us": sum(kept) / len(kept) / 1e3, "calls_in_run": len(v),
                      "avg_us_whole_run": sum(d for _, d in v) / len(v) / 1e3}
        total_us += sum(kept) / n_steps / 1e3
    # ---- PMC passes
    tot = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0, 0.0])
    by_kernel = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for ci, c in enumerate(("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES")):
        fs = glob.glob(f"{pmc_prefix}_{c}/**/*counter_collection.csv", recursive=True)
        if not fs:
            continue
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] != c:
                continue
            k = short_kernel_name(r["Kernel_Name"])
            e = tot[fam(k)]
            if ci < 2:
                e[ci] += float(r["Counter_Value"])
                by_kernel[k][ci] += float(r["Counter_Value"])
                if ci == 0:
                    e[2] += 1
                    by_kernel[k][2] += 1
            else:
                e[3] += float(r["Counter_Value"])
                e[4] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    lines = [f"Per kernel family, whole run of {pmc_steps} eager steps (bench.py's two arena-building steps included): fabric traffic = (2*FETCH_SIZE + WRITE_SIZE) KB,",
             "serialised kernel time and MFMA-busy cycles from the SQ_VALU_MFMA_BUSY_CYCLES pass; GB/s against the 8 TB/s HBM peak, MFMA busy =",
             "busy cycles / (time x 2.4 GHz x 1024 SIMDs).  Traffic includes infinity-cache hits (the counters sit at the L2 <-> fabric boundary)."]
    gt = tt = 0.0
    for k, (f, w, cnt, busy, dur) in sorted(tot.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1])):
        gb = (2 * f + w) * 1024 / 1e9
        gt += gb
        tt += dur
        gbs = gb / (dur * 1e-9) if dur else 0.0
        mf = busy / (dur * 1e-9 * 2.4e9 * 1024) if dur else 0.0
        lines.append(f"  {k:22s} read {2*f*1024/1e9:8.3f} GB  write {w*1024/1e9:8.3f} GB  time {dur/1e6:8.3f} ms  {gbs:7.0f} GB/s ({gbs/8000*100:4.1f} % of peak)  MFMA busy {mf*100:5.1f} %  launches {cnt:6d}")
    step_bytes = gt * 1e9 / pmc_steps if gt else None
    if gt:
        lines.append(f"  TOTAL {gt:.3f} GB in {tt/1e6:.3f} ms of kernel time = {gt/(tt*1e-9):.0f} GB/s")
        lines.append(f"  per step: {gt / pmc_steps:.3f} GB")
    open(os.path.join(out_dir, f"{tag}_step_traffic.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    for k, (f, w, n) in by_kernel.items():
        if k in kernels and n:
            kernels[k].update(hbm_bytes_per_launch=(2 * f + w) * 1024 / n, fetch_bytes_per_launch_corrected=2 * f * 1024 / n, write_bytes_per_launch=w * 1024 / n,
                              pmc_launches=n)
    lib = os.path.join(root, "focal_amd", "libfocal_hip.so")
    ref = {"note": "rocprofv3 --kernel-trace --stats -M of an eager `bench.py --no-graph` run (streams serialised by the profiler) + separate --pmc passes; "
                   "avg_us / calls_per_step over the steps after arena building and warm-up; hbm bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction, "
                   "MI355X_MICROARCH.md HBM section), counted at the L2 <-> fabric boundary (Infinity-Cache hits included).  Keys: tools/kernel_names.short_kernel_name "
                   "of the mangled symbol.  Only libfocal_hip's kernels count towards serialized_ms_per_step (torch's fills / copies are listed but not summed).",
           "workload": f"{model}/{dataset}", "steps_in_run": steps, "steps_skipped": skip,
           "lib_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16],
           "serialized_ms_per_step": sum(v["avg_us"] * v["calls_per_step"] for k, v in kernels.items() if not k.startswith(("__amd", "void at::", "at::"))) / 1e3,
           "serialized_ms_per_step_all_kernels": total_us / 1e3, "hbm_bytes_per_step": step_bytes, "kernels": kernels}
    json.dump(ref, open(os.path.join(out_dir, f"{tag}_reference.json"), "w"), indent=1)
    # ---- launch trace vs rocprofv3
    if dump and os.path.exists(dump):
        d = json.load(open(dump))
        rows = [f"libfocal_hip launch trace (bench.py --trace-dump, 5 eager one-stream steps, per-instance medians) vs rocprofv3 --kernel-trace on the same box, {model}/{dataset}",
                f"kernel time per step: rocprofv3 {ref['serialized_ms_per_step']:.3f} ms (library kernels), trace[dispatch] {d['dispatch']['kernel_ms_per_step']:.3f} ms, "
                f"trace[events] {d['events']['kernel_ms_per_step']:.3f} ms",
                f"{'kernel':70s} {'calls':>6s} {'rocprof us':>10s} {'dispatch':>9s} {'ratio':>6s} {'events':>9s} {'ratio':>6s}"]
        for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls_per_step"]):
            a, b = d["dispatch"]["kernels"].get(k), d["events"]["kernels"].get(k)
            if not a:
                continue
            rows.append(f"{k[:70]:70s} {v['calls_per_step']:6.1f} {v['avg_us']:10.2f} {a['avg_us']:9.2f} {a['avg_us']/v['avg_us']:6.3f} "
                        f"{(b or a)['avg_us']:9.2f} {(b or a)['avg_us']/v['avg_us']:6.3f}")
        open(os.path.join(out_dir, f"{tag}_trace_vs_rocprof.txt"), "w").write("\n".join(rows) + "\n")
        print("\n".join(rows[:40]))


if __name__ == "__main__":
    main()
