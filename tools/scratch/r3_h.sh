#!/bin/bash
for f in x i f if m s ms msf msif mf mif sif si sf; do echo "== flags $f"; timeout 60 python tools/scratch/graph_fork_nested.py $f 2>&1 | grep -v amdgpu | tail -1 | cut -c1-150; done
for p in 0 1; do echo "== thread prefork $p"; timeout 60 python tools/scratch/graph_fork_thread.py $p 2>&1 | grep -v amdgpu | tail -1 | cut -c1-150; done
