#!/bin/bash
# usage: bash tools/scratch/ab_env.sh "CFG_A" "CFG_B" ...   (each an env assignment list, "X=1" for the default) -- 3 interleaved repetitions
run() { env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 10 $BENCH_ARGS 2>&1 | tail -1 | sed 's/.*"value": \([0-9.]*\).*/\1/'; }
for rep in 1 2 3; do for cfg in "$@"; do echo "$rep | $cfg | $(run $cfg)"; done; done
