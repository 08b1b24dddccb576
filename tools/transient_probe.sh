#!/bin/bash
# the driver's command (--steps 20 --warmup 5) next to the builder's (--steps 50 --warmup 10) on ONE fresh box, with the
# per-step series (bench.py `step_series`) and a sysfs clock / power sampler beside them.  Output: gpurun_out/${tag}_*.json / .log
tag=${1:-r6_transient}
mkdir -p gpurun_out
python3 tools/smi_sampler.py 0.002 > gpurun_out/${tag}_smi.log 2>&1 &
SMI=$!
sleep 3
stamp() { python3 -c "import time; print('%.4f' % time.time(), '$1')" >> gpurun_out/${tag}_marks.log; }
for i in 1 2 3; do
  stamp "begin 20/5 #$i"
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-roofline > gpurun_out/${tag}_20_5_$i.json 2>> gpurun_out/${tag}.err
  stamp "end 20/5 #$i"
  sleep 4
  stamp "begin 50/10 #$i"
  python3 bench.py --gpus 1 --steps 50 --warmup 10 --no-secondary --no-cpu-baseline --no-roofline > gpurun_out/${tag}_50_10_$i.json 2>> gpurun_out/${tag}.err
  stamp "end 50/10 #$i"
  sleep 4
done
stamp "begin 200/0"
python3 bench.py --gpus 1 --steps 200 --warmup 0 --no-secondary --no-cpu-baseline --no-roofline > gpurun_out/${tag}_200_0.json 2>> gpurun_out/${tag}.err
stamp "end 200/0"
sleep 2
stamp "begin driver line"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_driver_line.json 2>> gpurun_out/${tag}.err
stamp "end driver line"
kill $SMI
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
