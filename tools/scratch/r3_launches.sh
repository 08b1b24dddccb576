#!/bin/bash
# new-kernel tests, parity suites, then same-box A/B of the launch-count changes
cd "$(dirname "$0")/../.."
python -m pytest tests/test_kernels_gpu.py -q -x -k "group or adamw or dropout" 2>&1 | tail -3
python -m pytest tests/test_swt_parity_gpu.py tests/test_4mod_gpu.py tests/test_train_dp_gpu.py -q -x 2>&1 | tail -3
bash tools/scratch/ab_env.sh "X=1" "FOCAL_NO_DW_RING_GROUP=1" "FOCAL_NO_MERGE_DW_GROUP=1" "FOCAL_NO_DW_RING_GROUP=1 FOCAL_NO_MERGE_DW_GROUP=1"
python bench.py --no-cpu-baseline --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['launches_per_step'], r['kernel'], r['frac'])"
