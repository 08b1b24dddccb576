#!/bin/bash
# same-box A/B: the persistent kernels' grids sized for fewer CUs than the chip has (FOCAL_LAB_CUS), the grouped weight gradients' workgroup target
run() { env "$@" python3 bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "default                         $(run X=1)"
  echo "FOCAL_LAB_CUS=224               $(run FOCAL_LAB_CUS=224)"
  echo "FOCAL_LAB_CUS=192               $(run FOCAL_LAB_CUS=192)"
  echo "FOCAL_LAB_DWG_TARGET=192        $(run FOCAL_LAB_DWG_TARGET=192)"
done
