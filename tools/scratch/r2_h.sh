#!/bin/bash
# round-2 experiment: asm LDS reads in the LDS-DMA kernels (real prefetch overlap) + 128x128 weight-gradient kernel
cd "$(dirname "$0")/../.."
out=gpurun_out/r2_h; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_swt_parity_gpu.py -m gpu -x -q > $out/tests.log 2>&1; tail -3 $out/tests.log
for v in "FOCAL_DW_NOWIDE=1" "FOCAL_DWW_VARIANT=0" "FOCAL_DWW_VARIANT=1" "FOCAL_DWW_VARIANT=2" "FOCAL_DWW_VARIANT=3" "FOCAL_DWW_VARIANT=0 FOCAL_DWW_WGS=256" "FOCAL_DWW_VARIANT=1 FOCAL_DWW_WGS=256"; do
  echo "== $v" >> $out/mb_dw.txt
  env $v timeout 300 python tools/mb_dw.py >> $out/mb_dw.txt 2>&1
done
for v in "FOCAL_PIPE_NST=2 FOCAL_DW_NOWIDE=1" "FOCAL_PIPE_NST=3 FOCAL_DW_NOWIDE=1" "FOCAL_PIPE_NST=4 FOCAL_DW_NOWIDE=1" "FOCAL_PIPE_NST=2" "FOCAL_PIPE_NST=3"; do
  echo "== $v" >> $out/bench.txt
  env $v timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200 >> $out/bench.txt
done
cat $out/bench.txt
