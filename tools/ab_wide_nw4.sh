#!/bin/bash
# same-box A/B: four consumer waves (64-row tiles) in the 256-channel one-launch MLP where they fit one round of the chip (the seismic encoder), against six everywhere
run() { env "$@" python3 bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "six consumer waves everywhere (default)                $(run X=1)"
  echo "four where 64-row tiles fit one round (FOCAL_LAB_WIDE_NW4=1)  $(run FOCAL_LAB_WIDE_NW4=1)"
done
