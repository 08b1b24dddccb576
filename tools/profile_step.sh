#!/bin/bash
# Usage (on the GPU box): bash tools/profile_step.sh <tag> [bench args...]
# rocprofv3 kernel-trace + stats of an eager (no hipGraph) bench run; the summary lands in gpurun_out/<tag>_kernel_stats.csv and
# the per-launch-shape table (tools/instance_table.py) in gpurun_out/<tag>_instances.csv
tag=$1; shift
root=$(pwd)
STEPS=10; WARM=3
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python3 $root/bench.py --no-graph --no-cpu-baseline --no-roofline --steps $STEPS --warmup $WARM "$@" > $root/gpurun_out/${tag}_bench.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $root/gpurun_out/${tag}_kernel_stats.csv
python3 $root/profiles/summarize.py $root/gpurun_out/${tag}_kernel_stats.csv $((STEPS+WARM+2)) 30
t=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 $root/tools/instance_table.py "$t" $((STEPS+WARM+2)) $root/gpurun_out/${tag}_instances.csv > $root/gpurun_out/${tag}_instances.txt
head -40 $root/gpurun_out/${tag}_instances.txt
[ -f $root/tools/copy_context.py ] && python3 $root/tools/copy_context.py "$t" > $root/gpurun_out/${tag}_copy_context.txt 2>&1
