#!/bin/bash
# same-box A/B: grouped weight gradients of 1 / 2 / 4 consecutive blocks per launch (FOCAL_DW_PAIR)
run() { env "$@" python3 bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "one block per launch  (FOCAL_DW_PAIR=1)  $(run FOCAL_DW_PAIR=1)"
  echo "two blocks per launch (FOCAL_DW_PAIR=2)  $(run FOCAL_DW_PAIR=2)"
  echo "four blocks per launch (FOCAL_DW_PAIR=4) $(run FOCAL_DW_PAIR=4)"
done
