#!/bin/bash
# same-box A/B of attention-backward prefetch depths: microbench (warm + cold) per stage geometry, then the full step
cd "$(dirname "$0")/../.."
for v in default attnd1 attnd2 attnd3; do
  lib=""; [ $v != default ] && lib="FOCAL_HIP_LIB=$(pwd)/focal_amd/lab/libfocal_hip_$v.so"
  echo "== $v warm"; env $lib python tools/microbench.py bf16 attn 2>&1 | grep attn_bwd
  echo "== $v cold"; env $lib FOCAL_MB_COLD=1 python tools/microbench.py bf16 attn 2>&1 | grep attn_bwd
done
python -m pytest tests/test_kernels_gpu.py -q -k attn 2>&1 | tail -2
BENCH_ARGS="" bash tools/scratch/ab_env.sh "X=1" "FOCAL_HIP_LIB=$(pwd)/focal_amd/lab/libfocal_hip_attnd1.so" "FOCAL_HIP_LIB=$(pwd)/focal_amd/lab/libfocal_hip_attnd2.so" "FOCAL_HIP_LIB=$(pwd)/focal_amd/lab/libfocal_hip_attnd3.so"
