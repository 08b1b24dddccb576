#!/bin/bash
# does a management-interface poll beside the bench (what a driver's GPU-busy sampler does) disturb replayed steps?
tag=${1:-r6_transient3}
mkdir -p gpurun_out
run() { python3 bench.py --gpus 1 --steps 200 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline > gpurun_out/${tag}_$1.json 2>> gpurun_out/${tag}.err; }
run quiet
( while true; do rocm-smi --showuse --showmemuse --json > /dev/null 2>&1; sleep 0.2; done ) &
P=$!
run rocmsmi_use
kill $P; wait $P 2>/dev/null
( while true; do rocm-smi -a --json > /dev/null 2>&1; sleep 0.2; done ) &
P=$!
run rocmsmi_all
kill $P; wait $P 2>/dev/null
( while true; do amd-smi metric --json > /dev/null 2>&1; sleep 0.2; done ) &
P=$!
run amdsmi_metric
kill $P; wait $P 2>/dev/null
( while true; do rocminfo > /dev/null 2>&1; sleep 0.2; done ) &
P=$!
run rocminfo
kill $P; wait $P 2>/dev/null
# CPU noise: 8 busy loops
for i in 1 2 3 4 5 6 7 8; do ( while true; do :; done ) & PIDS="$PIDS $!"; done
run cpu_noise8
kill $PIDS
python3 - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${tag}_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    s = d["step_series"]["timed_ms"]
    srt = sorted(s)
    print(f.split("/")[-1], d["value"], "ms/step", d["ms_per_step"], "median", srt[len(s)//2], "max", srt[-5:], "n>5.4:", sum(1 for v in s if v > 5.4))
    # worst 20-step window
    w = min(range(len(s)-19), key=lambda i: -sum(s[i:i+20]))
    print("   worst 20-step window: %.3f ms/step -> %.0f windows/s" % (sum(s[w:w+20])/20, 256/(sum(s[w:w+20])/20)*1e3))
PY
