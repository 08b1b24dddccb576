#!/bin/bash
# as transient_probe.sh but WITHOUT the sampler (reading hwmon every 2 ms may itself keep the SMU awake) and with long idle gaps
tag=${1:-r6_transient2}
gap=${2:-15}
mkdir -p gpurun_out
for i in 1 2 3; do
  sleep $gap
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline > gpurun_out/${tag}_20_5_$i.json 2>> gpurun_out/${tag}.err
  sleep $gap
  python3 bench.py --gpus 1 --steps 50 --warmup 10 --no-secondary --no-cpu-baseline --no-roofline > gpurun_out/${tag}_50_10_$i.json 2>> gpurun_out/${tag}.err
done
sleep $gap
rocm-smi --showclocks --showpower > gpurun_out/${tag}_idle_smi.txt 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_driver_line.json 2>> gpurun_out/${tag}.err
python3 - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${tag}_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    s = d["step_series"]
    print(f.split("/")[-1], d["value"], "ms/step", d["ms_per_step"])
    print("   warm ", s["warmup_ms"])
    print("   timed", s["timed_ms"][:60])
PY
