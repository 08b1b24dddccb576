#!/bin/bash
mkdir -p gpurun_out
{ echo "== deferred dW stream A/B (SW-T)"; bash tools/scratch/ab_env.sh "FOCAL_DW_STREAM=0" "X=1"; echo "== HAR4";  BENCH_ARGS="--dataset HAR4" bash tools/scratch/ab_env.sh "FOCAL_DW_STREAM=0" "X=1"; } 2>&1 | grep -v amdgpu > gpurun_out/r3_f.txt
python -m pytest tests/test_swt_parity_gpu.py tests/test_4mod_gpu.py tests/test_supervised_gpu.py tests/test_dp_parity_gpu.py tests/test_train_loop_gpu.py -q -x 2>&1 | tail -5 >> gpurun_out/r3_f.txt
cat gpurun_out/r3_f.txt
