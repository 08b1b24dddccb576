#!/bin/bash
# DeepSense counterpart of ablate.sh (FOCAL_ABLATE: timing-only diagnostic, results are garbage)
run() { env "$@" python3 bench.py --model DeepSense --no-cpu-baseline --no-roofline --no-secondary --steps 40 --warmup 10 2>&1 | tail -1 | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1/'; }
for cfg in "X=1" "FOCAL_ABLATE=gru_seq_fwd,gru_seq_bwd" "FOCAL_ABLATE=bn_stats" "FOCAL_ABLATE=bn_act_fwd" "FOCAL_ABLATE=bn_act_bwd" "FOCAL_ABLATE=conv_fwd" "FOCAL_ABLATE=conv_bwd_data" "FOCAL_ABLATE=conv_bwd_weight,linear_bwd_weight" "FOCAL_ABLATE=linear_fwd" "FOCAL_ABLATE=linear_bwd_data" "FOCAL_ABLATE=axpy,dropout" "FOCAL_ABLATE=packs" "X=2"; do
  echo "$cfg | $(run $cfg) ms | $(run $cfg) ms"
done
