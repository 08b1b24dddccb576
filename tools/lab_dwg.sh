#!/bin/bash
# lab: where the grouped weight-gradient launch's fixed cost goes -- no epilogue / plain stores / atomics, and the workgroup target
export FOCAL_MB_DW_BLOCKS="36864,128;18432,256;9216,256;4608,256"
for v in dwg_base dwg_store dwg_noepi; do
  echo "== $v"; FOCAL_HIP_LIB=focal_amd/lab/libfocal_hip_$v.so python3 tools/mb_dw.py group 2>&1 | grep "block rows"
done
for t in 128 192 512; do
  echo "== dwg_base target $t"; FOCAL_LAB_DWG_TARGET=$t FOCAL_HIP_LIB=focal_amd/lab/libfocal_hip_dwg_base.so python3 tools/mb_dw.py group 2>&1 | grep "block rows"
done
