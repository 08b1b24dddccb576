#!/bin/bash
# What does each kernel family contribute to the graph-replayed step WITH the real two-stream concurrency?  Each line is the step
# time with that family's launches skipped (FOCAL_ABLATE: timing diagnostic only, results are garbage); the difference to the
# full step is the family's marginal cost in situ, to be set against its serialised (profiler) kernel time.
run() { env "$@" python3 bench.py $ABL_ARGS --no-cpu-baseline --no-roofline --no-secondary --steps 40 --warmup 10 2>&1 | tail -1 | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1/'; }
for cfg in "X=1" "FOCAL_ABLATE=linear_bwd_weight" "FOCAL_ABLATE=linear_bwd_data" "FOCAL_ABLATE=linear_bwd_data_ln" "FOCAL_ABLATE=linear_fwd" "FOCAL_ABLATE=layernorm_bwd" "FOCAL_ABLATE=layernorm_fwd" "FOCAL_ABLATE=window_attn_bwd" "FOCAL_ABLATE=window_attn_fwd" "FOCAL_ABLATE=mlp_fwd" "FOCAL_ABLATE=mlp_bwd" "FOCAL_ABLATE=fft" "FOCAL_ABLATE=loss_head" "FOCAL_ABLATE=adamw" "FOCAL_ABLATE=embed" "X=2"; do
  echo "$cfg | $(run $cfg) ms | $(run $cfg) ms"
done
