#!/bin/bash
# same-box A/B of the one-launch MLP at 128 / 256 channels inside the replayed step (three interleaved repetitions)
run() { env "$@" python3 bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"; }
for i in 1 2 3; do
  echo "two launches each way (FOCAL_MLP_WIDE=0)          $(run FOCAL_MLP_WIDE=0)"
  echo "forward in one launch (FOCAL_MLP_WIDE_BWD=0)      $(run FOCAL_MLP_WIDE_BWD=0)"
  echo "forward and backward data path (default)          $(run X=1)"
done
