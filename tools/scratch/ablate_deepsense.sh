#!/bin/bash
# DeepSense counterpart of ablate.sh (FOCAL_ABLATE: timing-only diagnostic, results are garbage)
run() { env "$@" python bench.py --model DeepSense --no-cpu-baseline --no-roofline --steps 40 --warmup 10 2>&1 | tail -1 | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1/'; }
for cfg in "X=1" "FOCAL_ABLATE=gru_seq_fwd,gru_seq_bwd" "FOCAL_ABLATE=bn_stats,bn_act_fwd" "FOCAL_ABLATE=bn_act_bwd" "FOCAL_ABLATE=conv_fwd" "FOCAL_ABLATE=conv_bwd_data,conv_bwd_weight" "FOCAL_ABLATE=linear_fwd,linear_bwd_data,linear_bwd_weight" "X=2"; do
  echo "$cfg | $(run $cfg) ms | $(run $cfg) ms"
done
