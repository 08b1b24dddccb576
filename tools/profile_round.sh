#!/bin/bash
# Usage (on the GPU box): bash tools/profile_round.sh <tag> <model> <dataset> [extra bench args...]
# The round's evidence for one workload, all from the same box and library build:
#   (1) rocprofv3 --kernel-trace --stats -M of an eager bench run  -> gpurun_out/<tag>_kernel_stats.csv, <tag>_instances.txt
#   (2) --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES, one pass each (--kernel-trace only) -> <tag>_step_traffic.txt
#   (3) bench.py --trace-dump (the library's own launch trace, both modes) -> <tag>_trace_vs_rocprof.txt
#   (4) <tag>_reference.json: what bench.py's roofline is checked against (copy to profiles/r6_reference_<model>_<dataset>.json)
tag=$1; model=$2; dataset=$3; shift 3
root=$(pwd)
STEPS=10; WARM=3
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
# fixed step counts: bench.py's settle phase (untimed steps until three agree to 1 %) would add a run-dependent number of steps to every pass
export FOCAL_BENCH_NO_SETTLE=1
B="$root/bench.py --model $model --dataset $dataset --no-graph --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats -M --output-format csv -d /tmp/prof_$tag -o $tag -- python3 $B --steps $STEPS --warmup $WARM "$@" > $root/gpurun_out/${tag}_bench.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $root/gpurun_out/${tag}_kernel_stats.csv
t=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 $root/tools/instance_table.py "$t" $((STEPS+WARM+2)) $root/gpurun_out/${tag}_instances.csv > $root/gpurun_out/${tag}_instances.txt
PS=4; PW=2
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rocprofv3 --kernel-trace --pmc $c -M --output-format csv -d /tmp/pmcs_${tag}_$c -o $c -- python3 $B --steps $PS --warmup $PW "$@" > /tmp/pmcs_${tag}_$c.log 2>&1
done
python3 $root/bench.py --model $model --dataset $dataset --no-cpu-baseline --trace-dump $root/gpurun_out/${tag}_trace_dump.json "$@" > $root/gpurun_out/${tag}_trace_dump.log 2>&1
# the PMC runs have PS+PW+2 steps, the kernel trace STEPS+WARM+2: the reference takes durations from the trace and bytes per launch from the PMC passes
python3 $root/tools/rocprof_reference.py $tag $model $dataset $((STEPS+WARM+2)) $((WARM+2)) /tmp/prof_$tag /tmp/pmcs_$tag $((PS+PW+2)) $root/gpurun_out/${tag}_trace_dump.json
