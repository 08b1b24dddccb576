#!/bin/bash
# Usage: bash tools/ab_lib.sh <other library .so> [bench args]  -- same-box A/B of the built library against another build of it
# (FOCAL_HIP_LIB selects the library at run time), 3 interleaved repetitions
other=$1; shift
for i in 1 2 3; do
  r=$(python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary "$@" | python3 -c "import json,sys; p=json.loads(sys.stdin.readline()); print(p['value'], p['ms_per_step'])")
  echo "built library rep $i: $r"
  r=$(FOCAL_HIP_LIB=$other python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary "$@" | python3 -c "import json,sys; p=json.loads(sys.stdin.readline()); print(p['value'], p['ms_per_step'])")
  echo "$(basename $other) rep $i: $r"
done
