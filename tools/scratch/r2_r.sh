#!/bin/bash
# ring weight-gradient kernel: token slices per workgroup (FOCAL_DWR_HALVES = 1: 4 waves, one ring of 4 stages; 2: 8 waves, two rings of 4;
# 3: 12 waves, three rings of 3; 4: 16 waves, four rings of 2), reduction through LDS, one atomic pass per workgroup
cd "$(dirname "$0")/../.."
for h in 3 4; do FOCAL_DWR_HALVES=$h timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k linear_bwd 2>&1 | tail -1; done
for v in "FOCAL_DWR_HALVES=2" "FOCAL_DWR_HALVES=3" "FOCAL_DWR_HALVES=4"; do echo "== $v"; env $v python tools/mb_dw.py 2>&1 | grep -v amdgpu; done
bash tools/scratch/ab_env.sh "FOCAL_DWR_HALVES=2" "FOCAL_DWR_HALVES=3" "FOCAL_DWR_HALVES=4"
