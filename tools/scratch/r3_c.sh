#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -k "weight_gradient or linear_bwd" 2>&1 | tail -5 > gpurun_out/r3_c_dwgroup.txt
python tools/mb_dw.py group >> gpurun_out/r3_c_dwgroup.txt 2>&1
python tools/mb_dw.py >> gpurun_out/r3_c_dwgroup.txt 2>&1
{ echo "== grouped dW A/B (SW-T)"; bash tools/scratch/ab_env.sh "FOCAL_NO_DW_GROUP=1" "X=1"; echo "== HAR4"; BENCH_ARGS="--dataset HAR4" bash tools/scratch/ab_env.sh "FOCAL_NO_DW_GROUP=1" "X=1"; } >> gpurun_out/r3_c_dwgroup.txt 2>&1
{ echo "== test sequence"; python tools/scratch/dbg_fd_test_repro.py 25; echo "== test sequence, no streams"; FOCAL_NO_STREAMS=1 python tools/scratch/dbg_fd_test_repro.py 25; } > gpurun_out/r3_c_fd_repro.txt 2>&1
grep -v amdgpu.ids gpurun_out/r3_c_dwgroup.txt; grep -v amdgpu.ids gpurun_out/r3_c_fd_repro.txt | tail -30
