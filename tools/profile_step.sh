#!/bin/bash
# Usage (on the GPU box): bash tools/profile_step.sh <tag> [bench args...]
# rocprofv3 kernel-trace + stats of an eager (no hipGraph) bench run; the summary lands in gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
root=$(pwd)
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python3 $root/bench.py --no-graph --no-cpu-baseline --steps 10 --warmup 3 "$@" > $root/gpurun_out/${tag}_bench.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $root/gpurun_out/${tag}_kernel_stats.csv
python3 $root/profiles/summarize.py $root/gpurun_out/${tag}_kernel_stats.csv 13 30
