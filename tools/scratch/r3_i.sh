#!/bin/bash
mkdir -p gpurun_out
for args in "" "--model DeepSense" "--dataset HAR4" "--views random"; do
  python bench.py --no-cpu-baseline --steps 30 --warmup 8 $args 2>gpurun_out/err.log | tail -1 > gpurun_out/line.json || true
  python - "$args" <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/line.json"))
except Exception as e:
    print("FAILED", sys.argv[1], e); print(open("gpurun_out/err.log").read()[-2000:]); sys.exit(0)
r = d["roofline"]
print(f"== bench {sys.argv[1]!r}: {d['value']} windows/s, {d['ms_per_step']} ms/step; views: {d['config']['views'][:40]}")
print(f"   dominant: {r['kernel'][:90]}  frac {r['frac']} ({r['achieved']} {r['unit']}), {r['calls_per_step']} calls x {r['avg_us']} us = {r['ms_per_step']} ms; launches/step {r['launches_per_step']}, kernel ms/step {r['kernel_ms_per_step']}; traffic {r['traffic']} ({r['traffic_source']})")
for f in r["families"][:10]:
    print("   ", f)
PY
done
