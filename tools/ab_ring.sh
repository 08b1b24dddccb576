#!/bin/bash
# Same-box A/B of the ring GEMM dispatch: FOCAL_NO_RING unset / set, 3 interleaved repetitions.   bash tools/ab_ring.sh [bench args]
for i in 1 2 3; do
  r=$(python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary "$@" | python3 -c "import json,sys; p=json.loads(sys.stdin.readline()); print(p['value'], p['ms_per_step'])")
  echo "ring on  rep $i: $r"
  r=$(FOCAL_NO_RING=1 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary "$@" | python3 -c "import json,sys; p=json.loads(sys.stdin.readline()); print(p['value'], p['ms_per_step'])")
  echo "ring off rep $i: $r"
done
