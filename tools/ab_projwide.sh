#!/bin/bash
# same-box A/B: proj + residual + norm2 of the 128 / 256-channel blocks in front of the one-launch MLP (focal_mlp_wide_proj_fwd) against their own launches
run() { env "$@" python3 bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "fold at 64 only   (FOCAL_MLP_PROJ=64)       $(run FOCAL_MLP_PROJ=64)"
  echo "fold at 64,128    (FOCAL_MLP_PROJ=64,128)   $(run FOCAL_MLP_PROJ=64,128)"
  echo "fold at 64,256    (FOCAL_MLP_PROJ=64,256)   $(run FOCAL_MLP_PROJ=64,256)"
  echo "fold everywhere   (default)                 $(run X=1)"
done
