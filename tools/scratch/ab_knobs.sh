#!/bin/bash
# same-box sweep of launch-configuration knobs on the full step (3 repetitions each, interleaved)
run() { env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 10 2>&1 | tail -1 | sed 's/.*"value": \([0-9.]*\).*/\1/'; }
for rep in 1 2 3; do
  for cfg in "X=1" "FOCAL_GEMM_NOPIPE=1" "FOCAL_GEMM_PIPE_WIDE_N=384" "FOCAL_GEMM_PIPE_WIDE_N=128" "FOCAL_ATTN_BWD_NW=8" "FOCAL_ATTN_BWD_BLOCKS=512" "FOCAL_ATTN_BWD_NW=8 FOCAL_ATTN_BWD_BLOCKS=512" "FOCAL_DW_WGS=256" "FOCAL_DW_WGS=768" "FOCAL_EMBED_BLOCKS=256"; do
    echo "$rep | $cfg | $(run $cfg)"
  done
done
