#!/bin/bash
# same-box A/B: the 64-channel proj + residual + norm2 in front of the fused MLP, in its launch (focal_mlp_proj_fwd), against the two launches
run() { env "$@" python3 bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "proj as its own launch (FOCAL_MLP_PROJ=0)   $(run FOCAL_MLP_PROJ=0)"
  echo "proj in the MLP kernel (default)            $(run X=1)"
  echo "(bound: proj launches skipped)              $(run FOCAL_MLP_PROJ=0 FOCAL_ABLATE=proj64)"
done
