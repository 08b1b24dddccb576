#!/bin/bash
# same-box A/B: the next block's norm1 at 256 channels from the one-launch MLP's epilogue (bit-identical to the stand-alone ln_fwd launch)
run() { env "$@" python3 bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"; }
for i in 1 2 3 4; do
  echo "stand-alone ln_fwd at 256 (FOCAL_MLP_WIDE_LN256=0)   $(run FOCAL_MLP_WIDE_LN256=0)"
  echo "LayerNorm in the MLP kernel's epilogue (default)     $(run X=1)"
done
