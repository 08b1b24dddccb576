#!/bin/bash
run() { env "$@" python3 bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "full step                                   $(run X=1)"
  echo "(bound: stage-1/2 proj launches skipped)    $(run FOCAL_ABLATE=proj_wide)"
done
