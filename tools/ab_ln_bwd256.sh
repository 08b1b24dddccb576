#!/bin/bash
run() { env "$@" python3 bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"; }
for i in 1 2 3; do
  echo "LN backward in the dX epilogue up to 128 channels (default)   $(run FOCAL_LN_BWD_MAX_C=128)"
  echo "... up to 256 channels                                         $(run FOCAL_LN_BWD_MAX_C=256)"
done
