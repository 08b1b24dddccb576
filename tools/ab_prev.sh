#!/bin/bash
run() { env "$@" python3 bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"; }
for i in 1 2 3; do
  echo "previous build   $(run FOCAL_HIP_LIB=focal_amd/lab/libfocal_hip_prev.so)"
  echo "one-block loop   $(run X=1)"
done
