#!/bin/bash
run() { echo "== $*"; env "$@" PYTHONFAULTHANDLER=1 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>&1 | grep -v amdgpu | tail -2 | cut -c1-200; }
run X=1
run FOCAL_DW_KEEP_EVENTS=1
run FOCAL_DW_STREAM_DEPTH=100
run FOCAL_DW_STREAM_DEPTH=100 FOCAL_DW_KEEP_EVENTS=1
run FOCAL_DW_STREAM_DEPTH=4
