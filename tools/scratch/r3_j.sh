#!/bin/bash
mkdir -p gpurun_out
{ echo "== DeepSense: view passes ordered at the last BatchNorm (r2) vs side by side"; BENCH_ARGS="--model DeepSense" bash tools/scratch/ab_env.sh "FOCAL_DS_ORDERED_VIEWS=1" "X=1"; } 2>&1 | grep -v amdgpu > gpurun_out/r3_j.txt
python -m pytest tests/test_deepsense_parity_gpu.py tests/test_dp_parity_gpu.py tests/test_train_dp_gpu.py -q 2>&1 | tail -6 >> gpurun_out/r3_j.txt
cat gpurun_out/r3_j.txt
