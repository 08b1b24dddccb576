#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r2_i; mkdir -p $out
for v in "FOCAL_DWW_VARIANT=1 FOCAL_DWW_WGS=256" "FOCAL_DWW_VARIANT=1 FOCAL_DWW_WGS=256 FOCAL_DWW_NOBIAS=1" "FOCAL_DWW_VARIANT=2 FOCAL_DWW_WGS=256" "FOCAL_DWW_VARIANT=2 FOCAL_DWW_WGS=384"; do
  echo "== $v" >> $out/mb_dw.txt
  env $v timeout 300 python tools/mb_dw.py >> $out/mb_dw.txt 2>&1
done
grep -v amdgpu.ids $out/mb_dw.txt
