#!/bin/bash
run() { env "$@" python3 bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"; }
for i in 1 2 3; do
  echo "bwd off        $(run FOCAL_MLP_WIDE_BWD=0)"
  echo "bwd 128 only   $(run FOCAL_MLP_WIDE_BWD=128)"
  echo "bwd 256 only   $(run FOCAL_MLP_WIDE_BWD=256)"
  echo "bwd both       $(run X=1)"
done
