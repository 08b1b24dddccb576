#!/bin/bash
# round 3, first GPU call: fork-point A/B (forward passes of the modalities overlapping) + finite-difference outlier diagnosis
mkdir -p gpurun_out
{
echo "== SW_Transformer fork A/B"; bash tools/scratch/ab_env.sh "FOCAL_FORK_LATE=1" "X=1"
echo "== DeepSense fork A/B"; BENCH_ARGS="--model DeepSense" bash tools/scratch/ab_env.sh "FOCAL_FORK_LATE=1" "X=1"
echo "== HAR4 fork A/B"; BENCH_ARGS="--dataset HAR4" bash tools/scratch/ab_env.sh "FOCAL_FORK_LATE=1" "X=1"
} > gpurun_out/r3_a_fork_ab.txt 2>&1
{
echo "== default"; python tools/scratch/dbg_fd_outlier.py 30
echo "== no streams"; FOCAL_NO_STREAMS=1 python tools/scratch/dbg_fd_outlier.py 30
echo "== sync after update"; DBG_SYNC=1 python tools/scratch/dbg_fd_outlier.py 30
echo "== junk allocation, no update"; DBG_JUNK=1 python tools/scratch/dbg_fd_outlier.py 30
} > gpurun_out/r3_a_fd_outlier.txt 2>&1
tail -30 gpurun_out/r3_a_fork_ab.txt; tail -40 gpurun_out/r3_a_fd_outlier.txt
