#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_deepsense_parity_gpu.py tests/test_dp_parity_gpu.py tests/test_finetune_gpu.py tests/test_supervised_gpu.py -q -x 2>&1 | tail -4 > gpurun_out/r3_k.txt
{ echo "== DeepSense: weights re-ordered per pass (r2) vs once per step in one launch"; BENCH_ARGS="--model DeepSense" bash tools/scratch/ab_env.sh "FOCAL_DS_PACK_ONCE=0" "X=1"; } 2>&1 | grep -v amdgpu >> gpurun_out/r3_k.txt
cat gpurun_out/r3_k.txt
