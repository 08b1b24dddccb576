#!/bin/bash
# Soak on the final library: long replayed runs of the three workloads (rates flat, losses finite).  Usage (GPU box): bash tools/soak.sh
run() { python3 bench.py --no-cpu-baseline --no-roofline --no-secondary "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); t = d['step_series']['timed_ms']; n = len(t) // 4
q = [sum(t[i * n:(i + 1) * n]) / n for i in range(4)]
print(d['value'], 'windows/s', d['steps'], 'steps, mean ms per quarter', [round(x, 3) for x in q], 'max step', max(t), 'last loss', d['config'].get('last_loss'))"; }
echo "SW_Transformer, random views drawn on the device:  $(run --views random --steps 3000 --warmup 10)"
echo "DeepSense:                                        $(run --model DeepSense --steps 5000 --warmup 10)"
echo "SW_Transformer HAR4:                              $(run --dataset HAR4 --steps 1500 --warmup 10)"
