#!/bin/bash
# lab: what the dgamma / dbeta epilogue of the stand-alone LayerNorm backward costs (contended atomics vs plain stores vs nothing)
for v in "" focal_amd/lab/libfocal_hip_ln_noatomic.so focal_amd/lab/libfocal_hip_ln_noepi.so; do
  echo "== ${v:-shipped}"
  FOCAL_HIP_LIB=$v python3 tools/mb_ln_bwd.py
done
