#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -k "resid_ln or weight_gradient_group" 2>&1 | tail -5 > gpurun_out/r3_e.txt
python -m pytest tests/test_swt_parity_gpu.py tests/test_train_dp_gpu.py tests/test_4mod_gpu.py tests/test_supervised_gpu.py -q 2>&1 | tail -12 >> gpurun_out/r3_e.txt
{ echo "== wide LN fusion A/B (SW-T)"; bash tools/scratch/ab_env.sh "FOCAL_NO_LN_FUSE_WIDE=1" "X=1"; } >> gpurun_out/r3_e.txt 2>&1
cat gpurun_out/r3_e.txt; cat gpurun_out/observed_parity.json | grep -i "train_dp\|fd_dropout"
