#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_deepsense_parity_gpu.py tests/test_4mod_gpu.py -q -x 2>&1 | tail -4 > gpurun_out/r3_l.txt
python tools/grad_error_table.py SW_Transformer 256 2>&1 | grep -v amdgpu >> gpurun_out/r3_l.txt
python tools/grad_error_table.py DeepSense 256 2>&1 | grep -v amdgpu >> gpurun_out/r3_l.txt
cat gpurun_out/r3_l.txt | cut -c1-200
