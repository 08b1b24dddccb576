#!/bin/bash
mkdir -p gpurun_out
python tools/scratch/dbg_poison.py > gpurun_out/r3_d_poison.txt 2>&1
python -m pytest tests/test_train_dp_gpu.py tests/test_train_loop_gpu.py tests/test_host_paths_gpu.py tests/test_dp_parity_gpu.py -q 2>&1 | tail -15 > gpurun_out/r3_d_tests.txt
grep -v amdgpu.ids gpurun_out/r3_d_poison.txt | cut -c1-330; cat gpurun_out/r3_d_tests.txt
