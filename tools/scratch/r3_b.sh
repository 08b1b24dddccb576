#!/bin/bash
mkdir -p gpurun_out
{
for mode in item clone; do
echo "== $mode, fork late (round-2 behaviour)"; DBG_MODE=$mode FOCAL_FORK_LATE=1 python tools/scratch/dbg_fd_outlier.py 60
echo "== $mode, fork point"; DBG_MODE=$mode python tools/scratch/dbg_fd_outlier.py 60
echo "== $mode, fork late, no streams"; DBG_MODE=$mode FOCAL_FORK_LATE=1 FOCAL_NO_STREAMS=1 python tools/scratch/dbg_fd_outlier.py 60
echo "== $mode, fork late, sync after update"; DBG_MODE=$mode FOCAL_FORK_LATE=1 DBG_SYNC=1 python tools/scratch/dbg_fd_outlier.py 60
done
} > gpurun_out/r3_b_fd_outlier.txt 2>&1
grep -v amdgpu.ids gpurun_out/r3_b_fd_outlier.txt | tail -60
