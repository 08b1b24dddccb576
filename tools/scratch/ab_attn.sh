#!/bin/bash
# same-box A/B of the attention kernels: shipped library vs focal_amd/lab/libfocal_hip_attnold.so (the previous attn_mfma.hip)
cd "$(dirname "$0")/../.."
OLD=$(pwd)/focal_amd/lab/libfocal_hip_attnold.so
python -m pytest tests/test_kernels_gpu.py -q -k "attention" 2>&1 | tail -2
for v in new old; do
  lib=""; [ $v == old ] && lib="FOCAL_HIP_LIB=$OLD"
  echo "== $v warm"; env $lib python tools/microbench.py bf16 attn 2>&1 | grep "attn_"
  echo "== $v cold"; env $lib FOCAL_MB_COLD=1 python tools/microbench.py bf16 attn 2>&1 | grep "attn_"
done
bash tools/scratch/ab_env.sh "X=1" "FOCAL_HIP_LIB=$OLD"
