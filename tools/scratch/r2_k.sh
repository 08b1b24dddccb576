#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r2_k; mkdir -p $out
timeout 600 env FOCAL_DWR64=2 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "linear_bwd" > $out/tests.log 2>&1; tail -2 $out/tests.log
for v in "X=1" "FOCAL_DWR64=2" "FOCAL_DWR64=3" "FOCAL_DWR64=4" "FOCAL_DWR64=2 FOCAL_DW_WGS=768" "FOCAL_DWR64=3 FOCAL_DW_WGS=768"; do
  echo "== $v" >> $out/mb_dw.txt
  env $v timeout 300 python tools/mb_dw.py >> $out/mb_dw.txt 2>&1
done
grep -v amdgpu.ids $out/mb_dw.txt
bash tools/scratch/ab_env.sh "X=1" "FOCAL_DWR64=2" "FOCAL_DWR64=3" > $out/ab.txt 2>&1; cat $out/ab.txt
