#!/bin/bash
cd "$(dirname "$0")/../.."
FOCAL_MLP_BWD_SPLIT=1 timeout 600 python -m pytest tests/test_mlp_fused_gpu.py -m gpu -x -q 2>&1 | tail -3
for v in "X=1" "FOCAL_MLP_BWD_SPLIT=1"; do echo "== $v"; env $v python tools/mb_mlp.py 2>&1 | grep -v amdgpu | grep bwd; done
bash tools/scratch/ab_env.sh "X=1" "FOCAL_MLP_BWD_SPLIT=1"
