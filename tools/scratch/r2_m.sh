#!/bin/bash
# Row sweep of the weight-gradient kernels (profiles/r2_dw_fixed_cost.txt).  The middle run needs a lab build of the ring kernel whose
# epilogue stores nothing: add `#ifdef DW_RING_LAB_NO_ATOMICS` around the epilogue atomics of gemm_dw_ring.hpp, then
#   bash tools/scratch/build_variant.sh noatom "-DDW_RING_LAB_NO_ATOMICS" gemm_bf16.hip
cd "$(dirname "$0")/../.."
S="512,1024,256;2048,1024,256;4608,1024,256;9216,1024,256;18432,1024,256;36864,1024,256;4608,256,256;18432,256,256;73728,128,128;294912,64,64;8192,64,64"
echo "== ring kernel"; FOCAL_MB_DW_SHAPES="$S" python tools/mb_dw.py 2>&1 | grep -v amdgpu
if [ -f focal_amd/lab/libfocal_hip_noatom.so ]; then
  echo "== ring kernel without the atomic epilogue (lab build)"
  FOCAL_HIP_LIB=focal_amd/lab/libfocal_hip_noatom.so FOCAL_MB_DW_SHAPES="$S" python tools/mb_dw.py 2>&1 | grep -v amdgpu
fi
echo "== register-staged kernel"; FOCAL_DW_NORING=1 FOCAL_MB_DW_SHAPES="$S" python tools/mb_dw.py 2>&1 | grep -v amdgpu
