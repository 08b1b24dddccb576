#!/bin/bash
cd "$(dirname "$0")/../.."
python -m pytest tests/test_loss_gpu.py -q -x 2>&1 | tail -3
python -m pytest tests/test_swt_parity_gpu.py tests/test_dp_parity_gpu.py tests/test_train_dp_gpu.py -q -x 2>&1 | tail -2
OLD=$(pwd)/focal_amd/lab/libfocal_hip_lossold.so
echo "== new"; python tools/bench_loss_head.py 2>&1 | tail -8
echo "== old"; FOCAL_HIP_LIB=$OLD python tools/bench_loss_head.py 2>&1 | tail -8
bash tools/scratch/ab_env.sh "X=1" "FOCAL_HIP_LIB=$OLD"
BENCH_ARGS="--model DeepSense" bash tools/scratch/ab_env.sh "X=1" "FOCAL_HIP_LIB=$OLD"
