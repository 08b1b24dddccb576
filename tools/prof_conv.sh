#!/bin/bash
# kernel durations of the DeepSense convolution launches under rocprofv3 (the Python loop of tools/mb_conv.py is host-bound below ~18 us per call)
# usage: bash tools/prof_conv.sh [library ...]   ("" = the shipped library; FOCAL_CONV_RING=0 is always run last)
root=$(cd "$(dirname "$0")/.." && pwd)
run() {  # tag, env...
  tag=$1; shift
  cd /tmp && export TMPDIR=/tmp
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_conv_$tag -o conv -- python3 $root/tools/mb_conv.py > /tmp/prof_conv_$tag.log 2>&1
  cd $root
  echo "== $tag"
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/prof_conv_$tag/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    if "conv_ring" in r["Name"] or "focal_gemm_kernel" in r["Name"]:
        print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:7.1f} us  min {float(r["MinNs"]) / 1e3:7.1f}')
PY
}
run ring X=1
for lib in "$@"; do run $(basename $lib .so) FOCAL_HIP_LIB=$root/$lib; done
run gemm_path FOCAL_CONV_RING=0
