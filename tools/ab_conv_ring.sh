#!/bin/bash
# same-box A/B inside the DeepSense step: sliding-window GEMM / row-ring kernel / row-ring kernel + sums-only BatchNorm statistics (the default)
run() { env "$@" python3 bench.py --model DeepSense --no-cpu-baseline --no-roofline --no-secondary --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "sliding-window GEMM (FOCAL_CONV_RING=0)  $(run FOCAL_CONV_RING=0)"
  echo "row ring, one-launch statistics          $(run FOCAL_CONV_BN_SUMS=0)"
  echo "row ring, sums-only statistics (default) $(run X=1)"
done
