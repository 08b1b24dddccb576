#!/bin/bash
run() { env "$@" python3 bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"; }
for i in 1 2 3; do
  echo "default (6 / 6)      $(run X=1)"
  echo "256: 4 consumers     $(run FOCAL_HIP_LIB=focal_amd/lab/libfocal_hip_w256_4.so)"
  echo "128: 4 consumers     $(run FOCAL_HIP_LIB=focal_amd/lab/libfocal_hip_w128_4.so)"
  echo "128: 8 consumers     $(run FOCAL_HIP_LIB=focal_amd/lab/libfocal_hip_w128_8.so)"
done
