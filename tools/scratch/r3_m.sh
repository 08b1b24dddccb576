#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_dp_parity_gpu.py tests/test_train_dp_gpu.py tests/test_train_loop_gpu.py tests/test_swt_parity_gpu.py tests/test_supervised_gpu.py -q -x 2>&1 | tail -8 > gpurun_out/r3_m.txt
cat gpurun_out/r3_m.txt | cut -c1-600
