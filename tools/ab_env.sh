#!/bin/bash
# Usage: bash tools/ab_env.sh <ENVVAR> [bench args]  -- same-box A/B of one environment switch, 3 interleaved repetitions
v=$1; shift
for i in 1 2 3; do
  for on in 0 1; do
    r=$(env $v=$on python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline "$@" | python3 -c "import json,sys; p=json.loads(sys.stdin.readline()); print(p['value'], p['ms_per_step'])")
    echo "$v=$on rep $i: $r"
  done
done
